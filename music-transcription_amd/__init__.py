"""music_transcription_amd -- MI355X-native hot path of cs4247/music-transcription.

mel frontend -> CNN-RNN forward -> 88-pitch logits, as hand-written gfx950 kernels in
libmt_hip.so behind the reference's Python surface (TranscriptionModel, audio_to_mel).
Importing this package requires the built library; nothing here falls back to CPU code.
"""
from . import _lib                                   # noqa: F401  (raises if libmt_hip.so is missing)
from ._lib import MtError                            # noqa: F401
from .frontend import MelFrontend, audio_to_mel, get_frontend, mel_filterbank, num_frames   # noqa: F401
from .model import CNNRNNModel, CNNRNNModelLarge, TranscriptionModel                          # noqa: F401
from . import ops                                                                             # noqa: F401
from .ops import compute_loss, framewise_f1, mean_f1, predict_from_logits                     # noqa: F401
from .optim import FusedAdamClip, flatten_parameters, allreduce_mean_                        # noqa: F401
from .train import make_optimizer, train_one_epoch                                            # noqa: F401  (train.evaluate = validation loss)
from .data import CachedMaestroDataset, collate_fn, write_cache_chunk, write_cache_metadata   # noqa: F401

__version__ = "0.2.0"
