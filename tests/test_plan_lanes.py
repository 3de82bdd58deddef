"""C ABI of the launch lanes (csrc/plan.h): e2k_plan_lane / e2k_plan_event_record / e2k_plan_event_wait / e2k_plan_run_lanes on the
host model of the kernels -- argument checking, what gets recorded, and that a replay with one stream per lane, with one
stream for everything, and a profiled replay all reproduce the recorded calls."""
import ctypes

import pytest
import torch

bf16 = torch.bfloat16


def names(L, h):
    buf = ctypes.create_string_buffer(64)
    out = []
    for i in range(L.e2k_query_plan_size(h)):
        L.e2k_plan_op_name(h, i, ctypes.addressof(buf), 64)
        out.append(buf.value.decode())
    return out


def test_lane_calls_outside_a_recording_are_no_ops(emu):
    L = emu
    assert L.e2k_plan_lane(1) is None and L.e2k_plan_lane(0) is None
    L.e2k_plan_event_record(1, 0)
    L.e2k_plan_event_wait(0, 5)              # (not recording: nothing to check against)
    from e2_tts_pytorch_amd._lib import E2KError
    with pytest.raises(E2KError):
        L.e2k_plan_lane(7)                   # more lanes than the replay accepts streams for


def test_record_and_replay_with_lanes(emu):
    from e2_tts_pytorch_amd import ops
    from e2_tts_pytorch_amd._lib import E2KError
    L = emu
    torch.manual_seed(0)
    a = torch.randn(64, 64).to(bf16)
    b = torch.randn(32, 64).to(bf16)
    w = torch.randn(64, 32).to(bf16)
    y = torch.empty(64, 32, dtype=bf16)
    z = torch.empty(64, 64, dtype=bf16)
    scratch = torch.empty(16, dtype=torch.uint8)          # (a plan replays raw pointers: every buffer it touches must stay alive)
    lanes = ops.Lanes('cpu', [None, None])
    assert lanes.on and not lanes.cuda
    ops.begin_recording()
    try:
        with pytest.raises(E2KError):
            L.e2k_plan_event_wait(0, 3)      # waiting for an event nobody has recorded is a scheduling bug
        ops.gemm_nt(a, b, out=y)                             # lane 0
        e = lanes.record(ops.MAIN)
        lanes.wait(ops.TEXT, e)
        with lanes.lane(ops.TEXT):
            ops.gemm_nt(y, w, out=z)                         # lane 1, needs y
            with lanes.lane(ops.WGRAD):                      # nested: lane 2, back to lane 1 afterwards
                ops.fill_(scratch)
            assert lanes.cur == ops.TEXT
        lanes.join()
        h = ops.end_recording()
    except BaseException:
        ops.abort_recording()
        raise
    ns = names(L, h)
    assert ns == ['gemm_nt_bf16', 'lane_event_record', 'lane_event_wait', 'gemm_nt_bf16', 'fill_bytes',
                  'lane_event_record', 'lane_event_wait', 'lane_event_record', 'lane_event_wait'], ns
    assert [L.e2k_query_plan_op_lane(h, i) for i in range(len(ns))] == [0, 0, 1, 1, 2, 1, 0, 2, 0] and L.e2k_query_plan_op_lane(h, 99) == -1
    ref_y, ref_z = y.clone(), z.clone()
    arr3 = (ctypes.c_void_p * 3)()
    arr1 = (ctypes.c_void_p * 1)()
    for arr, n in ((arr3, 3), (arr1, 1)):
        y.zero_(); z.zero_()
        L.e2k_plan_run_lanes(h, 0, -1, ctypes.addressof(arr), n)
        assert torch.equal(y, ref_y) and torch.equal(z, ref_z)
    y.zero_(); z.zero_()
    L.e2k_plan_run(h, 0, -1, None)
    assert torch.equal(y, ref_y) and torch.equal(z, ref_z)
    ms = (ctypes.c_float * len(ns))()
    y.zero_(); z.zero_()
    L.e2k_plan_profile(h, 0, len(ns), ctypes.addressof(ms), None)
    assert torch.equal(z, ref_z) and all(m >= 0 for m in ms)
    with pytest.raises(E2KError):
        L.e2k_plan_run_lanes(h, 0, -1, ctypes.addressof(arr3), 9)
    # a segment that starts with a wait whose record lies in an earlier segment: events persist across calls of a plan
    L.e2k_plan_run_lanes(h, 0, 2, ctypes.addressof(arr3), 3)
    L.e2k_plan_run_lanes(h, 2, -1, ctypes.addressof(arr3), 3)
    assert torch.equal(z, ref_z)
    L.e2k_plan_free(h)
