"""ABI 110 on the device: (i) the size-prefixed descriptors run what the plain ones run; (ii) the tuning setters act on the calling host thread only -- two threads on
two streams force two different GEMM forms at the same time and each gets, bit for bit, what a single thread gets with that form."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tinychatengine_amd import capi
    capi.lib()
    return torch.device("cuda:0")


def _linear(dev, N, K, seed):
    from tinychatengine_amd.linear import Linear_half_int4
    g = torch.Generator(device=dev).manual_seed(seed)
    return Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack()


def test_size_prefixed_descriptors_run_the_plain_path(dev):
    from tinychatengine_amd import capi
    lin = _linear(dev, 1024, 1024, 5)
    g = torch.Generator(device=dev).manual_seed(6)
    for M in (1, 300):
        x = torch.empty(M, 1024, device=dev).normal_(0, 1, generator=g).to(torch.float16)
        y1 = torch.zeros(M, 1024, dtype=torch.float16, device=dev)
        y2 = torch.zeros_like(y1)
        y3 = torch.zeros_like(y1)
        st = torch.cuda.current_stream().cuda_stream
        capi.check(capi.w4a16_forward(lin.desc(x, y1), st))
        capi.check(capi.w4a16_forward_v2(lin.desc(x, y2), st))
        capi.check(capi.w4a16_forward_v2(lin.desc(x, y3), st, extra_zero_bytes=24))  # a later host's larger descriptor, its new fields zero
        torch.cuda.synchronize()
        assert torch.equal(y1, y2) and torch.equal(y1, y3) and bool(y1.float().abs().sum() > 0)
    A = torch.randint(-127, 128, (64, 256), device=dev, dtype=torch.int32).to(torch.int8)
    W = torch.randint(-127, 128, (128, 256), device=dev, dtype=torch.int32).to(torch.int8)
    b = torch.randint(-127, 128, (128,), device=dev, dtype=torch.int32).to(torch.int8)
    o1 = torch.zeros(64, 128, dtype=torch.int8, device=dev)
    o2 = torch.zeros_like(o1)
    mk = lambda o: capi.W8A8Desc(M=64, N=128, K=256, batch=1, A=A.data_ptr(), B=W.data_ptr(), bias=b.data_ptr(), C=o.data_ptr(), alpha=0.0005, beta=0.02, q_min=-128, q_max=127,
                                 bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8)
    capi.check(capi.w8a8_matmul(mk(o1), torch.cuda.current_stream().cuda_stream))
    capi.check(capi.w8a8_matmul_v2(mk(o2), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and bool(o1.float().abs().sum() > 0)


def test_two_host_threads_force_different_forms_at_once(dev):
    """Thread A forces the 128 x 128 form (mode 61), thread B the wide form's two quartets (2671, a different summation order: different bits) on the same linear and
    input, 40 launches each on its own stream, concurrently.  Each thread's every output equals the single-thread output of ITS form; the main thread's setting
    (the rule) is untouched."""
    from tinychatengine_amd import capi
    L = capi.lib()
    M, N, K = 512, 4096, 4096
    lin = _linear(dev, N, K, 9)
    g = torch.Generator(device=dev).manual_seed(10)
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)

    def run_form(mode, stream, n, sink):
        capi.check(L.tce_w4a16_set_debug_mode(mode))  # acts on THIS thread
        outs = []
        with torch.cuda.stream(stream):
            for _ in range(n):
                y = torch.zeros(M, N, dtype=torch.float16, device=dev)
                d = lin.desc(x, y)
                sink["what"] = capi.describe_dispatch(d)
                capi.check(capi.w4a16_forward(d, stream.cuda_stream))
                outs.append(y)
        stream.synchronize()
        sink["outs"] = outs
        capi.check(L.tce_w4a16_set_debug_mode(60))

    ref = {}
    for mode in (61, 2671):  # single-thread references, one form at a time
        s = {}
        run_form(mode, torch.cuda.Stream(), 1, s)
        ref[mode] = s["outs"][0]
    assert not torch.equal(ref[61], ref[2671]), "the two forms add their partial sums in different orders: they must differ somewhere, or the test shows nothing"
    sinks = {61: {}, 2671: {}}
    errs = []

    def worker(mode):
        try:
            run_form(mode, torch.cuda.Stream(), 40, sinks[mode])
        except Exception as e:  # noqa: BLE001
            errs.append((mode, repr(e)))

    ts = [threading.Thread(target=worker, args=(m,)) for m in (61, 2671)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert "quartets=1" in sinks[61]["what"] and "wave=128x64 quartets=2" in sinks[2671]["what"], (sinks[61]["what"], sinks[2671]["what"])
    for mode in (61, 2671):
        assert len(sinks[mode]["outs"]) == 40
        for y in sinks[mode]["outs"]:
            assert torch.equal(y, ref[mode]), f"thread forcing mode {mode}: an output differs from the single-thread run of that form"
    # the main thread never forced anything: its dispatch is the rule's
    y = torch.zeros(M, N, dtype=torch.float16, device=dev)
    assert "ksplit=2" in capi.describe_dispatch(lin.desc(x, y))
