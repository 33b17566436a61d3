"""Dependency-driven multi-stream executor for launch lists.

The reference leaves scheduling to TensorFlow's graph executor (inter-op parallelism over the Keras graph built at
facenet/models/inception_resnet_v1.py:482).  Here every launch declares the buffer regions it reads and writes;
``Schedule`` turns the program-order launch list into a placement on a few HIP streams plus the event edges that
preserve every RAW / WAR / WAW hazard.  Replayed eagerly it overlaps independent kernels (inception towers, wgrad
beside dgrad) on different hardware queues; captured into a HIP graph the same edges become graph dependencies.

Most launches of this network are latency-bound at < 1 block per CU (3x3 .. 17x17 maps), so concurrency between
independent launches is worth more than any single-kernel tuning (MI355X: 256 CUs, 8 XCDs).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

Region = Tuple[int, int, int]   # (base id, lo, hi): channel interval of an NHWC buffer or element interval of a flat one


@dataclass
class Op:
    name: str
    fn: Callable
    args: tuple
    keep: tuple = ()
    reads: Tuple[Region, ...] = ()
    writes: Tuple[Region, ...] = ()
    stream_hint: Optional[int] = None   # pin to a stream (collectives)


def region(t: torch.Tensor, lo: int = 0, hi: Optional[int] = None) -> Region:
    return (t.data_ptr(), lo, t.numel() if hi is None else hi)


class _Tracker:
    """last writer / readers-since per (base, interval); intervals of one base are identical, nested or disjoint."""

    def __init__(self):
        self.w: Dict[int, List[Tuple[int, int, int]]] = {}          # base -> [(lo, hi, op)]
        self.r: Dict[int, List[Tuple[int, int, int]]] = {}

    @staticmethod
    def _overlap(a_lo, a_hi, b_lo, b_hi):
        return a_lo < b_hi and b_lo < a_hi

    def deps_for_read(self, reg: Region) -> List[int]:
        b, lo, hi = reg
        return [op for (l, h, op) in self.w.get(b, []) if self._overlap(lo, hi, l, h)]

    def deps_for_write(self, reg: Region) -> List[int]:
        b, lo, hi = reg
        d = [op for (l, h, op) in self.w.get(b, []) if self._overlap(lo, hi, l, h)]
        d += [op for (l, h, op) in self.r.get(b, []) if self._overlap(lo, hi, l, h)]
        return d

    def note_read(self, reg: Region, op: int):
        self.r.setdefault(reg[0], []).append((reg[1], reg[2], op))

    def note_write(self, reg: Region, op: int):
        b, lo, hi = reg
        # a write supersedes covered entries; partially covered ones stay (conservative)
        self.w[b] = [(l, h, o) for (l, h, o) in self.w.get(b, []) if not (lo <= l and h <= hi)] + [(lo, hi, op)]
        self.r[b] = [(l, h, o) for (l, h, o) in self.r.get(b, []) if not (lo <= l and h <= hi)]


def levelize(ops: Sequence[Op]) -> List[int]:
    """ASAP dependency level of every launch (level = 1 + max level of what it depends on).  Launches of one level are
    mutually independent, so they may be reordered among themselves or fused into one grouped launch."""
    tr = _Tracker()
    level: List[int] = []
    for i, op in enumerate(ops):
        deps = set()
        for rg in op.reads:
            deps.update(tr.deps_for_read(rg))
        for rg in op.writes:
            deps.update(tr.deps_for_write(rg))
        deps.discard(i)
        level.append(1 + max((level[d] for d in deps), default=0))
        for rg in op.reads:
            tr.note_read(rg, i)
        for rg in op.writes:
            tr.note_write(rg, i)
    return level


class Schedule:
    """Placement of ``ops`` on ``n_streams`` streams.  steps = [('wait', stream, event) | ('run', stream, op_index) |
    ('record', stream, event)]; stream 0 is the caller's (capture) stream and everything is joined back to it."""

    def __init__(self, ops: Sequence[Op], n_streams: int = 4):
        self.ops = list(ops)
        self.n_streams = max(1, n_streams)
        self.steps: List[Tuple[str, int, int]] = []
        self.n_events = 0
        self._build()

    def _build(self):
        S = self.n_streams
        tr = _Tracker()
        op_stream: List[int] = []
        last_on_stream = [-1] * S                 # index of the last op placed on each stream
        seen: List[Dict[int, int]] = [dict() for _ in range(S)]   # stream s has synchronised with stream t up to op index
        ev_of_op: Dict[int, int] = {}
        for i, op in enumerate(self.ops):
            deps = set()
            for rg in op.reads:
                deps.update(tr.deps_for_read(rg))
            for rg in op.writes:
                deps.update(tr.deps_for_write(rg))
            deps.discard(i)
            if S == 1:
                s = 0
            elif op.stream_hint is not None:
                s = op.stream_hint % S
            else:
                # continue the chain of the most recent dependency if it is the tail of its stream
                tails = [d for d in deps if last_on_stream[op_stream[d]] == d]
                if tails:
                    s = op_stream[max(tails)]
                elif deps:
                    # dependencies are buried: take the stream that has been idle the longest
                    s = min(range(S), key=lambda k: last_on_stream[k])
                else:
                    s = 0
            for d in sorted(deps):
                t = op_stream[d]
                if t == s or seen[s].get(t, -1) >= d:
                    continue
                if d not in ev_of_op:            # record right after the producer (inserted retroactively below)
                    ev_of_op[d] = self.n_events
                    self.n_events += 1
                self.steps.append(("wait", s, ev_of_op[d]))
                seen[s][t] = d
            self.steps.append(("run", s, i))
            op_stream.append(s)
            last_on_stream[s] = i
            for rg in op.reads:
                tr.note_read(rg, i)
            for rg in op.writes:
                tr.note_write(rg, i)
        # insert the records directly after their producers
        out: List[Tuple[str, int, int]] = []
        for st in self.steps:
            out.append(st)
            if st[0] == "run" and st[2] in ev_of_op:
                out.append(("record", st[1], ev_of_op[st[2]]))
        # join every side stream back into stream 0
        for s in range(1, S):
            if last_on_stream[s] >= 0:
                e = self.n_events
                self.n_events += 1
                out.append(("record", s, e))
                out.append(("wait", 0, e))
        self.steps = out
        self.op_stream = op_stream

    def stats(self) -> Dict[str, int]:
        per = [0] * self.n_streams
        for s in self.op_stream:
            per[s] += 1
        return {"ops": len(self.ops), "events": self.n_events, **{f"stream{k}": v for k, v in enumerate(per)}}


class StreamSet:
    """Side streams of one device (created once)."""

    def __init__(self, device: torch.device, n_streams: int):
        self.device = device
        self.side = [torch.cuda.Stream(device=device) for _ in range(max(0, n_streams - 1))]


def make_events(sched: "Schedule") -> List[torch.cuda.Event]:
    """One event per edge (+1 for the initial fork).  A HIP-graph capture gets its own, never-reused set."""
    return [torch.cuda.Event() for _ in range(sched.n_events + 1)]


def run_schedule(sched: Schedule, streams: StreamSet, events: Optional[List[torch.cuda.Event]] = None):
    """Execute on the current stream (stream 0) and the side streams.  Safe under torch.cuda.graph capture: the side
    streams fork from and join into the capturing stream through the recorded events."""
    main = torch.cuda.current_stream(streams.device)
    ss = [main] + streams.side
    if events is None:
        if getattr(sched, "_events", None) is None:
            sched._events = make_events(sched)
        events = sched._events
    ev = events
    ops = sched.ops
    # side streams must not start before the caller's stream reaches this point
    if len(ss) > 1:
        e0 = ev[sched.n_events]
        e0.record(main)
        used = set(s for (k, s, _) in sched.steps if s > 0)
        for s in sorted(used):
            ss[s].wait_event(e0)
    ptrs = [s.cuda_stream for s in ss]
    for kind, s, x in sched.steps:
        if kind == "run":
            op = ops[x]
            rc = op.fn(*op.args, ptrs[s]) if not getattr(op.fn, "_torch_op", False) else _run_torch(op, ss[s])
            if rc:
                _lib.check(rc, op.name)
        elif kind == "record":
            ev[x].record(ss[s])
        else:
            ss[s].wait_event(ev[x])


def _run_torch(op: Op, stream: torch.cuda.Stream) -> int:
    with torch.cuda.stream(stream):
        op.fn(*op.args)
    return 0


def torch_op(fn: Callable) -> Callable:
    """Mark a Python callable that issues torch work on the CURRENT torch stream (zeroing, collectives)."""
    fn._torch_op = True
    return fn
