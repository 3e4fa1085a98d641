"""Standard MIDI File reader and piano-roll labels: what the reference gets from
`pretty_midi.PrettyMIDI(path).get_piano_roll(fs, times)[21:109] > 0` (data/dataset.py:133-148,:184-186).

pretty_midi (requirement `pretty_midi>=0.2.10`, unpinned) is not in /root/reference nor in this image, and the
reference holds no fixture for this boundary: PARITY UNPINNED.  This module restates the library's published
algorithm (pretty_midi.py `_load_tempo_changes` / `_load_instruments`, instrument.py `get_piano_roll`):
  * tick -> seconds through the tempo map of track 0 (default 120 bpm), resolution = ticks per beat;
  * a note-off (or note-on with velocity 0) closes every open note-on of its (channel, pitch) that started on an
    earlier tick; instruments are keyed by (program, channel, track); channel 9 is a drum track (empty roll);
  * per instrument: roll[pitch, int(start*fs):int(end*fs)] += velocity on a grid of int(fs*end_time) columns,
    sustain pedal (CC 64 >= 64) extends sounding notes by a running maximum over the pedalled span;
  * with `times`: column n = mean of the grid columns [round(times[n]*fs), round(times[n+1]*fs)) -- the LAST column
    is therefore always zero; instrument rolls are summed.
Pitch-bend events are read past (MAESTRO has none).  Host code: labels are tiny next to the audio side.
"""
from __future__ import annotations

import struct
from collections import defaultdict
from typing import Dict, List, Tuple

import numpy as np


class MidiError(ValueError):
    pass


def _read_vlq(b: bytes, i: int) -> Tuple[int, int]:
    v = 0
    while True:
        c = b[i]
        i += 1
        v = (v << 7) | (c & 0x7F)
        if not c & 0x80:
            return v, i


class Note:
    __slots__ = ("velocity", "pitch", "start", "end")

    def __init__(self, velocity, pitch, start, end):
        self.velocity, self.pitch, self.start, self.end = velocity, pitch, start, end

    def __repr__(self):
        return f"Note(start={self.start:.6f}, end={self.end:.6f}, pitch={self.pitch}, velocity={self.velocity})"


class Instrument:
    def __init__(self, program: int, is_drum: bool):
        self.program, self.is_drum = program, is_drum
        self.notes: List[Note] = []
        self.control_changes: List[Tuple[int, int, float]] = []      # (number, value, time)

    def get_end_time(self) -> float:
        ends = [n.end for n in self.notes] + [c[2] for c in self.control_changes]
        return max(ends) if ends else 0.0

    def get_piano_roll(self, fs: float = 100.0, times=None, pedal_threshold=64) -> np.ndarray:
        if not self.notes:
            return np.zeros((128, 0))
        roll = np.zeros((128, int(fs * self.get_end_time())))
        if self.is_drum:
            return roll if times is None else np.zeros((128, len(times)))
        for n in self.notes:
            roll[n.pitch, int(n.start * fs):int(n.end * fs)] += n.velocity
        if pedal_threshold is not None:
            t_on, on = 0, False
            for number, value, t in self.control_changes:
                if number != 64:
                    continue
                now = int(t * fs)
                cur = value >= pedal_threshold
                if not on and cur:
                    t_on, on = now, True
                elif on and not cur:
                    roll[:, t_on:now] = np.maximum.accumulate(roll[:, t_on:now], axis=1)
                    on = False
        if times is None:
            return roll
        idx = np.array(np.round(np.asarray(times, dtype=np.float64) * fs), dtype=np.int64)
        out = np.zeros((128, idx.shape[0]))
        for n, (s, e) in enumerate(zip(idx[:-1], idx[1:])):
            if s < roll.shape[1]:
                if s == e:
                    e = s + 1
                out[:, n] = roll[:, s:e].mean(axis=1)
        return out


class MidiFile:
    """Parsed SMF (format 0 or 1).  `.instruments`, `.resolution`, `.tick_to_time(tick)`."""

    def __init__(self, path_or_bytes):
        data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
        if data[:4] != b"MThd":
            raise MidiError("not a Standard MIDI File (no MThd)")
        hlen, fmt, ntrk, division = struct.unpack(">IHHH", data[4:14])
        if division & 0x8000:
            raise MidiError("SMPTE time division is not supported")
        self.resolution = division
        pos = 8 + hlen
        tracks: List[List[tuple]] = []
        for _ in range(ntrk):
            if data[pos:pos + 4] != b"MTrk":
                raise MidiError("missing MTrk chunk")
            (tlen,) = struct.unpack(">I", data[pos + 4:pos + 8])
            tracks.append(self._parse_track(data[pos + 8:pos + 8 + tlen]))
            pos += 8 + tlen
        self._build_tempo_map(tracks[0] if tracks else [])
        self._load_instruments(tracks)

    @staticmethod
    def _parse_track(b: bytes) -> List[tuple]:
        ev, i, tick, status = [], 0, 0, 0
        while i < len(b):
            d, i = _read_vlq(b, i)
            tick += d
            c = b[i]
            if c == 0xFF:                                   # meta
                mtype = b[i + 1]
                ln, j = _read_vlq(b, i + 2)
                if mtype == 0x51 and ln == 3:
                    ev.append((tick, "tempo", (b[j] << 16) | (b[j + 1] << 8) | b[j + 2]))
                i = j + ln
                if mtype == 0x2F:
                    break
                continue
            if c in (0xF0, 0xF7):                           # sysex
                ln, j = _read_vlq(b, i + 1)
                i = j + ln
                continue
            if c & 0x80:
                status = c
                i += 1
            elif not status:
                raise MidiError("running status without a status byte")
            kind, ch = status & 0xF0, status & 0x0F
            if kind in (0xC0, 0xD0):
                a = b[i]
                i += 1
                if kind == 0xC0:
                    ev.append((tick, "program", ch, a))
            else:
                a, v = b[i], b[i + 1]
                i += 2
                if kind == 0x90 and v > 0:
                    ev.append((tick, "on", ch, a, v))
                elif kind == 0x80 or kind == 0x90:
                    ev.append((tick, "off", ch, a))
                elif kind == 0xB0:
                    ev.append((tick, "cc", ch, a, v))
        return ev

    def _build_tempo_map(self, track0):
        changes = [(0, 500000)]                              # (tick, microseconds per beat): 120 bpm default
        for e in track0:
            if e[1] == "tempo":
                if e[0] == 0:
                    changes = [(0, e[2])]
                elif e[2] != changes[-1][1]:
                    changes.append((e[0], e[2]))
        ticks = np.array([c[0] for c in changes], dtype=np.int64)
        scale = np.array([c[1] * 1e-6 / self.resolution for c in changes])            # seconds per tick
        start = np.zeros(len(changes))
        for k in range(1, len(changes)):
            start[k] = start[k - 1] + (ticks[k] - ticks[k - 1]) * scale[k - 1]
        self._tempo_ticks, self._tempo_scale, self._tempo_start = ticks, scale, start

    def tick_to_time(self, tick: int) -> float:
        k = int(np.searchsorted(self._tempo_ticks, tick, side="right") - 1)
        return float(self._tempo_start[k] + (tick - self._tempo_ticks[k]) * self._tempo_scale[k])

    def _load_instruments(self, tracks):
        imap: Dict[tuple, Instrument] = {}
        stragglers: Dict[tuple, Instrument] = {}

        def get(program, channel, track, create):
            key = (program, channel, track)
            if key in imap:
                return imap[key]
            if create:
                inst = Instrument(program, channel == 9)
                if (channel, track) in stragglers:
                    inst.control_changes = stragglers[(channel, track)].control_changes
                imap[key] = inst
                return inst
            if (channel, track) not in stragglers:
                stragglers[(channel, track)] = Instrument(program, channel == 9)
            return stragglers[(channel, track)]

        for ti, track in enumerate(tracks):
            last_on = defaultdict(list)
            program = [0] * 16
            for e in track:
                tick, kind = e[0], e[1]
                if kind == "program":
                    program[e[2]] = e[3]
                elif kind == "on":
                    last_on[(e[2], e[3])].append((tick, e[4]))
                elif kind == "off":
                    key = (e[2], e[3])
                    if key in last_on:
                        opened = last_on[key]
                        close = [(s, v) for s, v in opened if s != tick]
                        keep = [(s, v) for s, v in opened if s == tick]
                        for s, v in close:
                            get(program[e[2]], e[2], ti, True).notes.append(Note(v, e[3], self.tick_to_time(s), self.tick_to_time(tick)))
                        if close and keep:
                            last_on[key] = keep
                        else:
                            del last_on[key]
                elif kind == "cc":
                    get(program[e[2]], e[2], ti, False).control_changes.append((e[3], e[4], self.tick_to_time(tick)))
        self.instruments = [i for i in imap.values()]

    def get_end_time(self) -> float:
        return max([i.get_end_time() for i in self.instruments], default=0.0)

    def get_piano_roll(self, fs: float = 100.0, times=None, pedal_threshold=64) -> np.ndarray:
        if not self.instruments:
            return np.zeros((128, 0))
        rolls = [i.get_piano_roll(fs=fs, times=times, pedal_threshold=pedal_threshold) for i in self.instruments]
        out = np.zeros((128, max(r.shape[1] for r in rolls)))
        for r in rolls:
            out[:, :r.shape[1]] += r
        return out


def chunk_roll(midi: MidiFile, start_time: float, end_time: float, sr: int = 16000, hop_length: int = 512) -> np.ndarray:
    """The reference's label for one chunk (data/dataset.py:134-148): (88, int((end-start)*fs)) float32 {0,1}."""
    fs = sr / hop_length
    times = np.linspace(start_time, end_time, int((end_time - start_time) * fs))
    return (midi.get_piano_roll(fs=fs, times=times)[21:109] > 0).astype(np.float32)
