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
    spans_n: Dict[int, int] = {}
    t0 = time.perf_counter()
    pending: List[torch.Tensor] = []             # decoded chunks not yet in a slab
    n_pending, n_slabs = 0, 0
    # Logits are held PER SLAB and only until every recording with chunks in the slab has been turned into notes: a slab's entry
    # is [logits (batch, 88, T), event behind its forward, recordings still to read it].
    slabs: List[list] = []
    segs: Dict[int, List[tuple]] = {}            # recording -> [(slab, first row, rows)] in chunk order
    open_recs: List[tuple] = []                  # (recording, first chunk of the shard, n chunks): chunks not yet all in launched slabs
    order: List[int] = []                        # recordings whose chunks are all in launched slabs, oldest first
    res: Dict[str, object] = {"chunks": 0, "notes": {}, "f1": {}, "n_notes": 0, "finite": True}
    finite_flags: List[torch.Tensor] = []
    fs = SR / HOP
    pos, cut = 0, 0                              # chunks decoded / chunks in launched slabs

    def launch(slab: torch.Tensor):
        nonlocal n_slabs, cut
        st = side[n_slabs % NS]
        st.wait_stream(main)                     # the slab was assembled on the main stream
        with torch.cuda.stream(st):
            mel, cmax = fe(slab, clamp=False)
            lg = net(mel, chunk_max_power=cmax)
            slab.record_stream(st)
            ev = torch.cuda.Event()
            ev.record(st)
        k, n = n_slabs, int(slab.shape[0])
        slabs.append([lg, ev, 0])
        # which recordings' chunks are rows [0, n) of this slab
        a = cut
        for rec in list(open_recs):
            i, first, cnt = rec
            lo, hi = max(first, a), min(first + cnt, a + n)
            if hi > lo:
                segs.setdefault(i, []).append((k, lo - a, hi - lo))
                slabs[k][2] += 1
            if first + cnt <= a + n:             # complete
                open_recs.remove(rec)
                order.append(i)
        cut += n
        n_slabs += 1

    f1_dev: Dict[int, torch.Tensor] = {}

    def finish(i):
        """Recording i's logits (its rows of the slabs it spans) -> notes / F1 on the device; the slabs are released behind it."""
        mine = segs.pop(i, [])
        if not mine:
            return
        parts = []
        for k, a, n in mine:
            main.wait_event(slabs[k][1])
            parts.append(slabs[k][0][a:a + n])
        lg = parts[0] if len(parts) == 1 else torch.cat(parts)
        finite_flags.append(torch.isfinite(lg).all())
        if want_notes:
            notes = tr.notes_from_logits_device(lg, threshold, fs)
            res["notes"][i] = notes
            res["n_notes"] += len(notes)
            path = midi_path_of(i) if midi_path_of else None
            if path:
                tr.write_midi(notes, path)
        if reference_roll_of is not None:
            ref = reference_roll_of(i, int(lg.shape[0]) * int(lg.shape[2]))
            if ref is not None:
                roll = predict_from_logits(lg, threshold).permute(1, 0, 2).reshape(88, -1)
                L = min(int(ref.shape[1]), int(roll.shape[1]))
                f1_dev[i] = framewise_f1(roll[None, :, :L].contiguous(), ref[None, :, :L].contiguous().float())
        del parts, lg
        for k, _, _ in mine:
            slabs[k][2] -= 1
            if slabs[k][2] == 0:
                slabs[k][0] = None               # (the caching allocator reuses the block for a later slab's logits)

    def drain(keep_in_flight: int):
        """Finish the recordings whose last slab has at least `keep_in_flight` younger slabs queued behind it (the host blocks on
        that slab's notes while the GPU still has the younger slabs to run)."""
        while order:
            i = order[0]
            last = max(k for k, _, _ in segs[i])
            if last > n_slabs - 1 - keep_in_flight:
                break
            order.pop(0)
            finish(i)

    for i in rec_ids:
        c = chunks_of(i)
        n_i = int(c.shape[0])
        spans_n[i] = n_i
        if n_i:
            open_recs.append((i, pos, n_i))
            pos += n_i
            pending.append(c)
            n_pending += n_i
        while n_pending >= batch:                # cut slabs off the front of the pending pool
            pool = pending[0] if len(pending) == 1 else torch.cat(pending)
            launch(pool[:batch])
            rest = pool[batch:]
            pending, n_pending = ([rest] if rest.shape[0] else []), int(rest.shape[0])
            drain(NS)
    if n_pending:
        launch(pending[0] if len(pending) == 1 else torch.cat(pending))
    drain(0)
    for st in side:
        main.wait_stream(st)
    res["chunks"] = pos
    res["f1"] = {i: float(v[0]) for i, v in f1_dev.items()}
    if finite_flags:
        res["finite"] = bool(torch.stack(finite_flags).all())
    torch.cuda.synchronize(dev)
    net.raise_on_handoff_timeout(sync=False)     # a timed-out recurrence leaves NaN logits = all-zero rolls: fail loudly
    res["wall_s"] = time.perf_counter() - t0
    res["slabs"] = n_slabs
    res["chunks_per_recording"] = {i: spans_n[i] for i in rec_ids}
    return res


class PcmSource:
    """chunks_of for transcribe_shard when the recordings are PCM frames in (pinned) HOST memory, as a WAV file holds them:
    (frames, channels) int16 / int32 / float32 at `rate`.  Recording order is known up front, so the H2D copy of recording
    k + `ahead` is queued on a copy stream while recording k is resampled (mt_resample_polyphase: channel mean + PCM scaling +
    polyphase filter in one kernel) and cut into zero-padded 30 s chunks (main.py:60-100) -- librosa.load(sr=16000, mono=True) +
    split_audio_into_chunks of the reference, on the device."""

    def __init__(self, pcm_of: Callable[[int], torch.Tensor], rate_of: Callable[[int], int], rec_ids: Sequence[int], device, ahead: int = 2):
        self.pcm_of, self.rate_of, self.ids, self.dev, self.ahead = pcm_of, rate_of, list(rec_ids), torch.device(device), max(0, ahead)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.index = {i: k for k, i in enumerate(self.ids)}
        self.inflight: Dict[int, tuple] = {}
        self.cursor = 0
        self.bytes_h2d = 0

    def _prefetch(self, upto: int):
        while self.cursor < min(upto + 1, len(self.ids)):
            i = self.ids[self.cursor]
            host = self.pcm_of(i)
            with torch.cuda.stream(self.copy_stream):
                d = host.to(self.dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self.bytes_h2d += host.numel() * host.element_size()
            self.inflight[i] = (d, ev)
            self.cursor += 1

    def __call__(self, i: int) -> torch.Tensor:
        self._prefetch(self.index[i] + self.ahead)
        d, ev = self.inflight.pop(i)
        main = torch.cuda.current_stream(self.dev)
        main.wait_event(ev)
        d.record_stream(main)
        y = tr.resample_pcm_device(d, self.rate_of(i), SR)
        return tr.split_into_chunks_device(y)[0]


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
