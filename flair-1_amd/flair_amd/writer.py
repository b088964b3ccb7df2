"""Mirror of /root/reference/src/flair/writer.py (``predictionwriter`` :8-68) with batched, asynchronous
device-to-host traffic.

The reference moves int64 predictions to the host (8 bytes/pixel), casts to uint8 and LZW-encodes tile by tile on
the training thread.  Here the uint8 cast happens on the device, the copy lands in a pinned staging buffer on a
side stream (1 byte/pixel), and encoding runs on a worker thread while the next batch is on the GPU.
``georeferencing_output`` needs rasterio (absent in this image): that branch raises; the plain branch writes the
same ``PRED_<name>`` LZW TIFFs through PIL as the reference does (writer.py:45-50).
"""
from __future__ import annotations

import atexit
import queue
import threading
import weakref
from pathlib import Path

import torch

try:  # pragma: no cover - lightning is absent in the build image
    from pytorch_lightning.callbacks import BasePredictionWriter as _Base
except Exception:  # noqa: BLE001
    class _Base:  # minimal stand-in with the attribute the hooks read
        def __init__(self, write_interval="batch"):
            self.interval = type("Interval", (), {"on_batch": write_interval in ("batch", "batch_and_epoch")})()


def pred_filename(output_dir: str, filename: str) -> str:
    return str(output_dir + "/" + "PRED_" + filename.split("/")[-1])


class predictionwriter(_Base):
    def __init__(self, config, output_dir, write_interval, max_pending: int = 4):
        super().__init__(write_interval)
        self.config = config
        self.output_dir = output_dir
        Path(self.output_dir).mkdir(exist_ok=True, parents=True)
        self._q: queue.Queue = queue.Queue(maxsize=max_pending)
        self._worker = None
        self._error = None
        self._copy_stream = None

    # -- worker: waits for the copy, then encodes
    def _run(self):
        from PIL import Image
        while True:
            item = self._q.get()
            if item is None:
                return
            host, done, filenames = item
            try:
                done.synchronize()
                arr = host.numpy()
                for prediction, filename in zip(arr, filenames):
                    Image.fromarray(prediction).save(pred_filename(self.output_dir, filename), compression="tiff_lzw")
            except Exception as e:  # noqa: BLE001
                self._error = e
            finally:
                self._q.task_done()

    def write_on_batch_end(self, trainer, pl_module, prediction, batch_indices, batch, batch_idx, dataloader_idx):
        if self.config["georeferencing_output"]:
            raise RuntimeError("georeferencing_output needs rasterio, which this build does not ship; "
                               "set georeferencing_output: False")
        if self._error is not None:
            raise self._error
        preds, filenames = prediction["preds"], prediction["id"]
        if not preds.is_cuda:
            raise RuntimeError("predictionwriter expects device predictions (no CPU path)")
        if self._worker is None:
            # A daemon thread + an atexit hook that drains the queue: tiles still queued when the interpreter exits reach the
            # disk, and an exit without close() cannot hang.  (A non-daemon worker is joined by threading._shutdown() BEFORE
            # the atexit callbacks run, i.e. while it still blocks in q.get(): the stopper never fired.)
            self._worker = threading.Thread(target=self._run, daemon=True)
            self._worker.start()
            self._copy_stream = torch.cuda.Stream(device=preds.device)
            ref = weakref.ref(self)
            atexit.register(lambda: (ref() is not None) and ref()._close_quietly())
        u8 = preds if preds.dtype == torch.uint8 else preds.to(torch.uint8)  # astype('uint8'), on the device
        host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
        self._copy_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._copy_stream):
            host.copy_(u8, non_blocking=True)
            u8.record_stream(self._copy_stream)
            done = torch.cuda.Event()
            done.record()
        self._q.put((host, done, list(filenames)))

    def on_predict_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        if not self.interval.on_batch:
            return
        batch_indices = getattr(getattr(trainer, "predict_loop", None), "current_batch_indices", None)
        self.write_on_batch_end(trainer, pl_module, outputs, batch_indices, batch, batch_idx, dataloader_idx)

    # Lightning calls these at the end of trainer.predict(); the reference reads the PRED_* files right after it
    # (main.py:238-242), so every queued tile must be on disk and a worker-side failure must surface here.
    def on_predict_epoch_end(self, trainer=None, pl_module=None, *args, **kwargs):
        self.flush()

    def on_predict_end(self, trainer=None, pl_module=None):
        self.close()

    def teardown(self, trainer=None, pl_module=None, stage=None):
        self.close()

    def flush(self):
        """Block until every queued tile is on disk; re-raise what the worker caught."""
        if self._worker is not None:
            self._q.join()
        if self._error is not None:
            err, self._error = self._error, None
            raise err

    def _stop_worker(self):
        if self._worker is not None:
            self._q.join()
            self._q.put(None)
            self._worker.join()
            self._worker = None

    def close(self):
        self._stop_worker()
        if self._error is not None:
            err, self._error = self._error, None
            raise err

    def _close_quietly(self):
        try:
            self._stop_worker()
        except Exception:  # noqa: BLE001
            pass
