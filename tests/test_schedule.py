"""Host logic of the dependency scheduler (no GPU): every RAW / WAR / WAW hazard of the program-order launch list must
be ordered in the multi-stream placement, either by stream order or through a record -> wait event chain."""
import random

import pytest

from facenet_amd.schedule import Op, Schedule


def _conflicts(a: Op, b: Op):
    def ov(x, y):
        return x[0] == y[0] and x[1] < y[2] and y[1] < x[2]
    for w in a.writes:
        if any(ov(w, r) for r in b.reads) or any(ov(w, w2) for w2 in b.writes):
            return True
    return any(ov(r, w) for r in a.reads for w in b.writes)


def _happens_before(sched: Schedule):
    """Simulate the step list with vector clocks; returns clock[i][s] = ops on stream s known complete before op i starts."""
    S = sched.n_streams
    clock = [[-1] * S for _ in range(S)]      # per stream: latest op index of every stream it has synchronised with
    ev = {}
    before = {}
    for kind, s, x in sched.steps:
        if kind == "run":
            before[x] = list(clock[s])
            clock[s][s] = x
        elif kind == "record":
            ev[x] = list(clock[s])
        else:
            assert x in ev, "wait on an event that was never recorded earlier in issue order"
            clock[s] = [max(a, b) for a, b in zip(clock[s], ev[x])]
    return before


def _random_program(n_ops, n_bufs, seed):
    rnd = random.Random(seed)
    ops = []
    for i in range(n_ops):
        def reg():
            b = rnd.randrange(n_bufs)
            lo = rnd.choice([0, 0, 32, 64])
            return (b, lo, lo + rnd.choice([32, 64, 128]))
        ops.append(Op(f"op{i}", None, (), reads=tuple(reg() for _ in range(rnd.randrange(0, 3))),
                      writes=tuple(reg() for _ in range(rnd.randrange(1, 3)))))
    return ops


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("streams", [1, 2, 4])
def test_all_hazards_are_ordered(seed, streams):
    ops = _random_program(120, 9, seed)
    sched = Schedule(ops, streams)
    before = _happens_before(sched)
    assert sorted(x for k, _, x in sched.steps if k == "run") == list(range(len(ops)))
    for j in range(len(ops)):
        for i in range(j):
            if _conflicts(ops[i], ops[j]):
                si = sched.op_stream[i]
                assert before[j][si] >= i, f"op{i} (stream {si}) and op{j} (stream {sched.op_stream[j]}) conflict but are unordered"


def test_independent_chains_spread_over_streams():
    # three towers reading one trunk, then a join: the towers must not all land on one stream
    trunk = (1, 0, 256)
    ops = [Op("produce", None, (), writes=(trunk,))]
    for t in range(3):
        ops.append(Op(f"a{t}", None, (), reads=(trunk,), writes=((10 + t, 0, 32),)))
        ops.append(Op(f"b{t}", None, (), reads=((10 + t, 0, 32),), writes=((20, 32 * t, 32 * t + 32),)))
    ops.append(Op("join", None, (), reads=((20, 0, 96),), writes=((30, 0, 256),)))
    s = Schedule(ops, 4)
    towers = {s.op_stream[1 + 2 * t] for t in range(3)}
    assert len(towers) == 3
    assert all(s.op_stream[1 + 2 * t] == s.op_stream[2 + 2 * t] for t in range(3))     # a chain stays on its stream
    st = s.stats()
    assert st["ops"] == len(ops) and st["events"] >= 3
    # every side stream is joined back into stream 0 at the end
    tail = s.steps[-2 * (len(towers) - 1):]
    assert all(k in ("record", "wait") for k, _, _ in tail)


def test_single_stream_is_program_order():
    ops = _random_program(40, 5, 3)
    s = Schedule(ops, 1)
    assert [x for k, _, x in s.steps if k == "run"] == list(range(40))
    assert s.n_events == 0


def test_levelize_gives_a_valid_topological_order():
    from facenet_amd.schedule import levelize
    ops = _random_program(150, 8, 11)
    lv = levelize(ops)
    for j in range(len(ops)):
        for i in range(j):
            if _conflicts(ops[i], ops[j]):
                assert lv[j] > lv[i]                      # dependent launches are on strictly later levels
    order = sorted(range(len(ops)), key=lambda i: (lv[i], i))
    pos = {i: k for k, i in enumerate(order)}
    for j in range(len(ops)):
        for i in range(j):
            if _conflicts(ops[i], ops[j]):
                assert pos[i] < pos[j]
