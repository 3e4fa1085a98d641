"""music_transcription_amd -- MI355X-native hot path of cs4247/music-transcription.

mel frontend -> CNN-RNN forward -> 88-pitch logits, as hand-written gfx950 kernels in
libmt_hip.so behind the reference's Python surface (TranscriptionModel, audio_to_mel).
Importing this package requires the built library; nothing here falls back to CPU code.
"""
import os as _os
import warnings as _warnings


def _request_hw_queues(n: int = 8) -> None:
    """The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise); two
    streams on one queue run their kernels one after the other.  The multi-stream schedules of this package (4 forwards in
    flight on 4 streams next to torch's own: transcribe_chunks*, scripts/transcribe_corpus.py, bench.py) need one queue per
    stream: measured 8 750 chunks/s with 4 queues against 10 490 with 8, same kernels.  The variable is read when the runtime
    initialises, so it is set here, at import, when the caller has not chosen a value; if the runtime is already up in this
    process with fewer queues, say so instead of silently losing the overlap (a reference script that swaps its import for
    this package, INTEGRATION.md section 1, gets the setting without touching its own code)."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return
    import sys as _sys
    _torch = _sys.modules.get("torch")
    if _torch is not None and _torch.cuda.is_initialized():
        _warnings.warn(f"music_transcription_amd: the HIP runtime was initialised before this import without GPU_MAX_HW_QUEUES; "
                       f"streams beyond the default 4 hardware queues will serialise (export GPU_MAX_HW_QUEUES={n} before the first "
                       f"GPU call to keep multi-stream throughput)", RuntimeWarning, stacklevel=3)
        return
    _os.environ["GPU_MAX_HW_QUEUES"] = str(n)


_request_hw_queues()

from . import _lib                                   # noqa: F401,E402  (raises if libmt_hip.so is missing)
from ._lib import MtError                            # noqa: F401
from .frontend import MelFrontend, audio_to_mel, get_frontend, mel_filterbank, num_frames   # noqa: F401
from .model import CNNRNNModel, CNNRNNModelLarge, TranscriptionModel                          # noqa: F401
from . import ops                                                                             # noqa: F401
from .ops import compute_loss, framewise_f1, mean_f1, predict_from_logits                     # noqa: F401
from .optim import FusedAdamClip, flatten_parameters, allreduce_mean_                        # noqa: F401
from .train import make_optimizer, train_one_epoch                                            # noqa: F401  (train.evaluate = validation loss)
from .data import CachedMaestroDataset, collate_fn, write_cache_chunk, write_cache_metadata   # noqa: F401

__version__ = "0.3.0"
