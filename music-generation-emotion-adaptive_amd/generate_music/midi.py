"""Token list -> notes -> Standard MIDI File bytes, without pretty_midi (absent offline).

Mirrors the detokeniser loop of the reference endpoint (api_cache.py:208-221): an `[INSTRUMENT] X`
token opens an instrument (General-MIDI program of X, 0 when unknown); a
`[NOTE] [PITCH:p] [START:s] [END:e] [DURATION:d]` token (api_cache.py:157 `note_re`) adds a note with
velocity 100 to the current instrument; everything else is ignored.  Host-side string work.

Pitch names follow music21, the library that produced the training tokens
(midi_test/midi_extract.py): `-` is a flat (`B-4` = B flat 4), `#` a sharp.
"""
from __future__ import annotations

import re
import struct
from dataclasses import dataclass, field
from typing import List, Sequence

note_re = re.compile(r"\[NOTE\] \[PITCH:(.+?)\] \[START:(.+?)\] \[END:(.+?)\] \[DURATION:(.+?)\]")

# the General MIDI programs the reference's FAMILY_TO_INSTRUMENTS can produce (api_cache.py:152-156)
GM_PROGRAMS = {"Acoustic Grand Piano": 0, "Violin": 40, "Flute": 73, "Acoustic Guitar (nylon)": 24, "Trumpet": 56,
               "Acoustic Bass": 32, "Synth Bass 1": 38, "Cello": 42, "Clarinet": 71}
_PC = {"C": 0, "D": 2, "E": 4, "F": 5, "G": 7, "A": 9, "B": 11}


def note_name_to_number(name: str) -> int:
    m = re.match(r"^([A-Ga-g])([#\-b]*)(-?\d+)$", name.strip())
    if not m:
        raise ValueError(f"Improper note format: {name}")
    pc = _PC[m.group(1).upper()] + m.group(2).count("#") - m.group(2).count("-") - m.group(2).count("b")
    return 12 * (int(m.group(3)) + 1) + pc


@dataclass
class Note:
    pitch: int
    start: float
    end: float
    velocity: int = 100


@dataclass
class Instrument:
    program: int
    name: str
    notes: List[Note] = field(default_factory=list)


def tokens_to_instruments(tokens: Sequence[str]) -> List[Instrument]:
    instruments: List[Instrument] = []
    current = None
    for tok in tokens:
        if tok.startswith("[INSTRUMENT]"):
            name = tok.split("]", 1)[1].strip()
            current = Instrument(GM_PROGRAMS.get(name, 0), name)
            instruments.append(current)
        elif current is not None:
            m = note_re.match(tok)
            if m:
                current.notes.append(Note(note_name_to_number(m.group(1)), float(m.group(2)), float(m.group(3))))
    return instruments


def _vlq(n: int) -> bytes:
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def write_midi(instruments: Sequence[Instrument], tempo_bpm: float = 120.0, resolution: int = 220) -> bytes:
    """Format-1 SMF: a tempo track + one track per instrument (channel i, skipping the drum channel)."""
    ticks = lambda sec: int(round(sec * resolution * tempo_bpm / 60.0))
    tracks = [b"\x00\xff\x51\x03" + struct.pack(">I", int(round(60e6 / tempo_bpm)))[1:] + b"\x00\xff\x2f\x00"]
    for i, inst in enumerate(instruments):
        ch = i % 15
        ch = ch + 1 if ch >= 9 else ch
        events = []
        for n in inst.notes:
            p = min(127, max(0, n.pitch))
            events.append((ticks(n.start), 1, bytes([0x90 | ch, p, n.velocity & 0x7F])))
            events.append((ticks(max(n.end, n.start)), 0, bytes([0x80 | ch, p, 0])))
        events.sort(key=lambda e: (e[0], e[1]))
        name = inst.name.encode("utf-8")[:127]
        data = b"\x00\xff\x03" + bytes([len(name)]) + name + b"\x00" + bytes([0xC0 | ch, inst.program & 0x7F])
        t = 0
        for tick, _, msg in events:
            data += _vlq(tick - t) + msg
            t = tick
        tracks.append(data + b"\x00\xff\x2f\x00")
    out = b"MThd" + struct.pack(">IHHH", 6, 1, len(tracks), resolution)
    for tr in tracks:
        out += b"MTrk" + struct.pack(">I", len(tr)) + tr
    return out


def tokens_to_midi(tokens: Sequence[str]) -> bytes:
    return write_midi(tokens_to_instruments(tokens))
