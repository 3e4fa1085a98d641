"""Offline transcription of a whole corpus on ONE rank's shard (BASELINE.json configs[4]).

The reference transcribes a recording as a Python loop over its 30 s chunks, batch 1 (main.py:229-362).  Chunks keep no
cross-chunk state (main.py:258-266), so here they are batched ACROSS recordings: the shard's chunks are streamed in slabs of
`batch` chunks -- a slab is assembled as soon as enough recordings have been decoded, never the whole shard first -- and slab f
runs mel + forward on stream f % streams.  Per recording the device then turns its logits into notes (threshold + chunk
concatenation + per-pitch run-length: mt_roll_to_notes, main.py:153-226), so what crosses PCIe per recording is its note list
(two ints per note), not its piano roll; framewise F1 against a reference roll, when one is given, is computed on the device too.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import transcribe as tr
from .frontend import get_frontend
from .ops import framewise_f1, predict_from_logits

SR, CH, HOP = 16000, 480000, 512


@torch.no_grad()
def transcribe_shard(model, rec_ids: Sequence[int], chunks_of: Callable[[int], torch.Tensor], *, n_mels: int, device,
                     batch: int = 128, streams: int = 3, threshold: float = 0.5, want_notes: bool = True,
                     reference_roll_of: Optional[Callable[[int, int], Optional[torch.Tensor]]] = None,
                     midi_path_of: Optional[Callable[[int], Optional[str]]] = None, warm: bool = True) -> Dict[str, object]:
    """rec_ids: the recordings of this rank; chunks_of(i) -> (n_i, 480000) float32 CUDA tensor (decode + resample + split for
    real audio: part of the measured time; a view of resident synthetic audio otherwise).  Returns {"wall_s", "chunks",
    "notes": {i: [(pitch, start, end)]}, "f1": {i: float}, "n_notes", "finite"}; wall_s covers slab assembly, every forward,
    the note extraction and the F1 counts (one device synchronisation at the end)."""
    dev = torch.device(device)
    net = model.model
    fe = get_frontend(SR, n_mels, HOP, str(dev))
    NS = max(1, streams)
    side = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    main = torch.cuda.current_stream(dev)
    if warm:                                     # weight packing, code objects, every stream's workspace: not part of wall_s
        w0 = torch.zeros(batch, CH, device=dev)
        torch.cuda.synchronize(dev)
        for st in side:
            with torch.cuda.stream(st):
                m0, c0 = fe(w0, clamp=False)
                net(m0, chunk_max_power=c0)
        torch.cuda.synchronize(dev)
        del w0
    t0 = time.perf_counter()
    pending: List[torch.Tensor] = []             # decoded chunks not yet in a slab
    n_pending, n_slabs = 0, 0
    outs: List[torch.Tensor] = []                # logits per slab, in chunk order
    spans = {}                                   # recording -> (first chunk, n chunks) in shard order
    pos = 0

    def launch(slab: torch.Tensor):
        nonlocal n_slabs
        st = side[n_slabs % NS]
        st.wait_stream(main)                     # the slab was assembled on the main stream
        with torch.cuda.stream(st):
            mel, cmax = fe(slab, clamp=False)
            outs.append(net(mel, chunk_max_power=cmax))
            slab.record_stream(st)
        n_slabs += 1

    for i in rec_ids:
        c = chunks_of(i)
        spans[i] = (pos, int(c.shape[0]))
        pos += int(c.shape[0])
        pending.append(c)
        n_pending += int(c.shape[0])
        while n_pending >= batch:                # cut slabs off the front of the pending pool
            pool = pending[0] if len(pending) == 1 else torch.cat(pending)
            launch(pool[:batch])
            rest = pool[batch:]
            pending, n_pending = ([rest] if rest.shape[0] else []), int(rest.shape[0])
    if n_pending:
        launch(pending[0] if len(pending) == 1 else torch.cat(pending))
    for st in side:
        main.wait_stream(st)
    n_chunks = pos
    res: Dict[str, object] = {"chunks": n_chunks, "notes": {}, "f1": {}, "n_notes": 0, "finite": True}
    if n_chunks:
        logits = outs[0] if len(outs) == 1 else torch.cat(outs)           # (n_chunks, 88, T): 0.33 MB per chunk, stays on the GPU
        res["finite"] = bool(torch.isfinite(logits).all())
        fs = SR / HOP
        for i in rec_ids:
            a, n = spans[i]
            if n == 0:
                continue
            if want_notes:
                notes = tr.notes_from_logits_device(logits[a:a + n], threshold, fs)
                res["notes"][i] = notes
                res["n_notes"] += len(notes)
                path = midi_path_of(i) if midi_path_of else None
                if path:
                    tr.write_midi(notes, path)
            if reference_roll_of is not None:
                T_total = n * logits.shape[2]
                ref = reference_roll_of(i, T_total)
                if ref is not None:
                    roll = predict_from_logits(logits[a:a + n], threshold).permute(1, 0, 2).reshape(88, -1)
                    L = min(int(ref.shape[1]), int(roll.shape[1]))
                    res["f1"][i] = float(framewise_f1(roll[None, :, :L].contiguous(), ref[None, :, :L].contiguous().float())[0])
    torch.cuda.synchronize(dev)
    net.raise_on_handoff_timeout(sync=False)     # a timed-out recurrence leaves NaN logits = all-zero rolls: fail loudly
    res["wall_s"] = time.perf_counter() - t0
    res["slabs"] = n_slabs
    res["chunks_per_recording"] = {i: spans[i][1] for i in rec_ids}
    return res


def synthetic_corpus(n_recordings: int, hours: float, seed: int = 0):
    """Durations (s) of a MAESTRO-test-like corpus: gamma(2.5)-distributed lengths scaled to `hours` in total."""
    import numpy as np
    rng = np.random.default_rng(seed)
    raw = rng.gamma(2.5, 1.0, size=n_recordings)
    return list(raw / raw.sum() * hours * 3600.0)


def synth_recording(i: int, seconds: float, device, seed: int = 0) -> torch.Tensor:
    """Noise + a few decaying partials, generated on the GPU; the last chunk zero-padded in the waveform domain (main.py:93-95)."""
    n = int(seconds * SR)
    g = torch.Generator(device=device).manual_seed(seed * 100003 + i)
    nch = max(1, -(-n // CH))
    t = torch.arange(nch * CH, device=device, dtype=torch.float32) / SR
    y = 0.1 * torch.randn(nch * CH, device=device, generator=g)
    for k in range(4):
        f0 = 27.5 * 2.0 ** (float(torch.randint(0, 88, (1,), device=device, generator=g)) / 12.0)
        y += 0.3 * torch.exp(-((t * (0.5 + k)) % 3.0)) * torch.sin(2 * torch.pi * f0 * t)
    y[n:] = 0.0
    return y.clamp_(-1, 1).view(nch, CH)
