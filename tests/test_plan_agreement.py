"""CPU tests of what keeps an N-rank frame from hanging (host logic only: frame plans are host data,
the in-process communicator's control plane is memory): avr_comm_control_allgather and
avr_frame_plan_agree with N rank THREADS -- ranks whose plans describe one exchange agree; a rank
handed another camera, another ownership or other settings makes EVERY rank return the error; a
rank that never arrives ends its peers' wait at the deadline; ranks in different collectives are
told apart.  (The reference finds a mismatch in the metadata message of every transfer,
Common/Image.cpp:62-90, and its waits complete or error, DirectSendBase.cpp:206-220, 277.)"""
import ctypes as C
import threading
import time

import pytest

from amrvolumerenderer_amd import _capi, scenes
from amrvolumerenderer_amd.compositor import FramePlan
from amrvolumerenderer_amd.types import make_params


def local_comms(lib, n):
    handles = (C.c_void_p * n)()
    _capi.check(lib.avr_comm_create_local(n, handles))
    return [C.c_void_p(h) for h in handles]


def run_ranks(n, body):
    """body(rank) on n threads (ctypes releases the GIL inside the C ABI); returns results / errors."""
    out = [None] * n

    def work(rank):
        try:
            out[rank] = ("ok", body(rank))
        except Exception as error:  # noqa: BLE001 -- what the rank's call raised is the result
            out[rank] = ("error", str(error))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert all(not t.is_alive() for t in threads)
    return out


@pytest.fixture()
def scene():
    spec = scenes.make_amr_scene(32, 2, 8, "smooth")
    scenes.assign_owners(spec, 3, "level_pairs")
    boxes = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    params = make_params(75, 43, spec.scalar_range, 0.8, 0.01, spec.bounds)
    return spec, boxes, params


def test_control_allgather_between_rank_threads(avr_lib):
    n = 4
    comms = local_comms(avr_lib, n)

    def body(rank):
        mine = bytes([rank + 1] * 16)
        everybody = C.create_string_buffer(16 * n)
        _capi.check(avr_lib.avr_comm_control_allgather(comms[rank], None, mine, everybody, 16))
        return everybody.raw

    got = run_ranks(n, body)
    want = b"".join(bytes([r + 1] * 16) for r in range(n))
    assert got == [("ok", want)] * n
    assert [avr_lib.avr_comm_control_rounds(c) for c in comms] == [1] * n
    with pytest.raises(ValueError):    # a multiple of 4, at most AVR_CONTROL_MAX_BYTES
        _capi.check(avr_lib.avr_comm_control_allgather(comms[0], None, b"abc", C.create_string_buffer(12), 3))
    for c in comms:
        avr_lib.avr_comm_destroy(c)


def test_plans_of_one_frame_agree_and_others_do_not(avr_lib, scene):
    spec, boxes, params = scene
    n = 3
    comms = local_comms(avr_lib, n)
    cam, other = scenes.default_camera(), scenes.orbit_camera(5)
    plans = [FramePlan(boxes, params, cam, r, n) for r in range(n)]

    def agree(per_rank_plans, digests=(7, 7, 7)):
        return run_ranks(n, lambda r: _capi.check(avr_lib.avr_frame_plan_agree(
            per_rank_plans[r]._handle, comms[r], None, digests[r])))

    assert agree(plans) == [("ok", None)] * n
    # rank 1 was handed another camera: every rank returns the error, and the same one
    mixed = [plans[0], FramePlan(boxes, params, other, 1, n), plans[2]]
    got = agree(mixed)
    assert all(kind == "error" and "rank 1's plan differs from rank 0's" in text for kind, text in got), got
    # ... other settings (the frame driver mixes its settings into the digest)
    got = agree(plans, digests=(7, 7, 8))
    assert all(kind == "error" and "rank 2's plan differs" in text for kind, text in got), got
    # ... another ownership of the boxes on one rank
    moved = [scenes.metadata_box(spec, i) for i in range(len(spec.boxes))]
    far = next(i for i, b in enumerate(moved) if b.owner != moved[0].owner)
    moved[0].owner, moved[far].owner = moved[far].owner, moved[0].owner
    got = agree([FramePlan(moved, params, cam, 0, n), plans[1], plans[2]])
    assert all(kind == "error" and "differs from rank 0's" in text for kind, text in got), got
    # a plan made for another rank count is refused outright
    with pytest.raises(ValueError):
        _capi.check(avr_lib.avr_frame_plan_agree(FramePlan(boxes, params, cam, 0, 2)._handle, comms[0],
                                                 None, 0))
    for c in comms:
        avr_lib.avr_comm_destroy(c)


def test_a_rank_that_never_arrives_ends_in_an_error_at_the_deadline(avr_lib, scene):
    _, boxes, params = scene
    n = 3
    comms = local_comms(avr_lib, n)
    plans = [FramePlan(boxes, params, scenes.default_camera(), r, n) for r in range(n)]
    _capi.check(avr_lib.avr_set_frame_timeout_ms(400))
    try:
        begin = time.monotonic()

        def body(rank):
            if rank == 2:
                return "absent"       # this rank never calls
            _capi.check(avr_lib.avr_frame_plan_agree(plans[rank]._handle, comms[rank], None, 0))

        got = run_ranks(n, body)
        elapsed = time.monotonic() - begin
        assert got[2] == ("ok", "absent")
        assert all(kind == "error" and "AVR_FRAME_TIMEOUT_MS" in text for kind, text in got[:2]), got
        assert 0.3 < elapsed < 5.0
        # the communicator is broken for good: a later meeting fails at once, for the late rank too
        begin = time.monotonic()
        with pytest.raises(_capi.AvrError):
            _capi.check(avr_lib.avr_frame_plan_agree(plans[2]._handle, comms[2], None, 0))
        assert time.monotonic() - begin < 0.2
    finally:
        _capi.check(avr_lib.avr_set_frame_timeout_ms(-1))
        for c in comms:
            avr_lib.avr_comm_destroy(c)


def test_ranks_in_different_control_rounds_are_told_apart(avr_lib):
    """A rank that sends 16 bytes while its peers send 8 is not in the same round: every rank gets the
    error (over RCCL such rounds would never end: the deadline ends those)."""
    n = 3
    comms = local_comms(avr_lib, n)

    def body(rank):
        size = 16 if rank == 1 else 8
        _capi.check(avr_lib.avr_comm_control_allgather(comms[rank], None, bytes(size),
                                                       C.create_string_buffer(size * n), size))

    got = run_ranks(n, body)
    assert all(kind == "error" and "the ranks' calls differ" in text for kind, text in got), got
    for c in comms:
        avr_lib.avr_comm_destroy(c)


def test_a_callers_allgather_that_hangs_ends_at_the_deadline(avr_lib):
    """avr_comm_set_control: the caller's own allgather (MPI in the reference's host, gloo in
    bench.py) runs under the deadline of every other wait of a frame -- a peer that never arrives
    inside THAT collective must not leave the rank waiting for as long as the caller's library does
    (gloo: 30 minutes).  The callback is run on a helper thread over copies of the buffers; past the
    deadline the call returns the error (and the helper is left behind)."""
    comms = local_comms(avr_lib, 2)
    release = threading.Event()
    calls = []

    @_capi.CONTROL_ALLGATHER_FN
    def allgather(_user, mine, out, size):
        calls.append(size)
        if len(calls) == 1:   # the first round works: every rank "says" what this one says
            C.memmove(out, mine, size)
            C.memmove(out + size, mine, size)
            return 0
        release.wait(20)       # the second one hangs (a peer died inside the collective)
        return 0

    _capi.check(avr_lib.avr_comm_set_control(comms[0], allgather, None))
    everybody = C.create_string_buffer(16)
    _capi.check(avr_lib.avr_comm_control_allgather(comms[0], None, b"12345678", everybody, 8))
    assert everybody.raw == b"1234567812345678"
    _capi.check(avr_lib.avr_set_frame_timeout_ms(300))
    try:
        begin = time.monotonic()
        with pytest.raises(_capi.AvrError) as failure:
            _capi.check(avr_lib.avr_comm_control_allgather(comms[0], None, b"12345678", everybody, 8))
        assert "AVR_FRAME_TIMEOUT_MS" in str(failure.value) and "caller's allgather" in str(failure.value)
        assert 0.25 < time.monotonic() - begin < 3.0
    finally:
        release.set()          # (lets the helper thread go before the callback object dies)
        time.sleep(0.2)
        _capi.check(avr_lib.avr_set_frame_timeout_ms(-1))
        for c in comms:
            avr_lib.avr_comm_destroy(c)
