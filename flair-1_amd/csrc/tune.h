// Diagnostic tuning switches: one integer per key, default = environment variable of the same name (read once), else the
// built-in default; flair_tune_set() overrides it at run time so that kernel variants can be A/B-timed inside ONE process
// (cdna_hip_programming.md §5.4 rule 24).  Not part of the reference boundary.
#pragma once

namespace flair {
int tune(const char* key, int dflt);       // current value of the switch
void tune_set(const char* key, int value);
}  // namespace flair
