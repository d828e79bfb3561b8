"""vc_lstm_persist_fwd_noalloc (the entry point that allocates nothing: its exchange buffer is an array inside the
library) against vc_lstm_persist_fwd (caller-supplied exchange buffer), both called through the C ABI with ctypes on
the same inputs.  Two calls in a row share the library's one buffer per GPU, the second with a smaller input, and a
shape beyond the buffer's size is refused.  The recurrence is deterministic (fixed summation order), so the bar is
bit equality of h, c and the saved gate activations."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _arr(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _inputs(T, B, H, ndir, seed, dev):
    g = torch.Generator().manual_seed(seed)
    gx = [torch.randn(T, B, 4 * H, generator=g).to(dev) for _ in range(ndir)]
    whh = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dev).to(torch.bfloat16) for _ in range(ndir)]
    return gx, whh


def _outputs(T, B, H, ndir, dev):
    # NaN-filled: every element must be written by the call
    hs = torch.full((T, B, H * ndir), float("nan"), device=dev, dtype=torch.bfloat16)
    act = [torch.full((T, B, 4 * H), float("nan"), device=dev) for _ in range(ndir)]
    c = [torch.full((T, B, H), float("nan"), device=dev) for _ in range(ndir)]
    return hs, act, c


def _run(lib, noalloc, gx, whh, T, B, H, ndir, dev):
    hs, act, c = _outputs(T, B, H, ndir, dev)
    err = torch.zeros(1, device=dev, dtype=torch.int32)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    hs_ptrs = (ctypes.c_void_p * ndir)(*[hs.data_ptr() + d * H * 2 for d in range(ndir)])
    if noalloc:
        rc = lib.vc_lstm_persist_fwd_noalloc(1, _arr(gx), _arr(whh), hs_ptrs, H * ndir, _arr(act), _arr(c),
                                             ctypes.c_void_p(err.data_ptr()), T, B, H, ndir, stream)
    else:
        lib.vc_lstm_persist_xchg_bytes.restype = ctypes.c_longlong
        nb = lib.vc_lstm_persist_xchg_bytes(B, H)
        xch = [torch.empty(nb, device=dev, dtype=torch.uint8) for _ in range(ndir)]
        rc = lib.vc_lstm_persist_fwd(1, _arr(gx), _arr(whh), hs_ptrs, H * ndir, _arr(act), _arr(c), _arr(xch),
                                     ctypes.c_void_p(err.data_ptr()), T, B, H, ndir, stream)
    torch.cuda.synchronize()
    lib.vc_last_error.restype = ctypes.c_char_p
    assert rc == 0, lib.vc_last_error()
    assert int(err.item()) == 0, "a workgroup timed out"
    return hs, act, c


def _same(a, b):
    ha, aa, ca = a
    hb, ab, cb = b
    assert not torch.isnan(ha.float()).any()
    assert torch.equal(ha.view(torch.int16), hb.view(torch.int16))
    for x, y in zip(aa + ca, ab + cb):
        assert not torch.isnan(x).any()
        assert torch.equal(x, y)


def test_noalloc_entry_matches_caller_buffer_entry(cuda):
    from video_classification_amd import _lib
    lib = _lib.lib()
    # first call: the headline decoder's shape, both directions (256 KiB of the 512 KiB buffer)
    T, B, H, ndir = 16, 32, 512, 2
    gx, whh = _inputs(T, B, H, ndir, 1, cuda)
    ref = _run(lib, False, gx, whh, T, B, H, ndir, cuda)
    got = _run(lib, True, gx, whh, T, B, H, ndir, cuda)
    _same(got, ref)
    # second call right after it, smaller in every extent: step tags left by the first call must not be taken for
    # this call's
    T2, B2, H2, nd2 = 7, 5, 64, 1
    gx2, whh2 = _inputs(T2, B2, H2, nd2, 2, cuda)
    got2 = _run(lib, True, gx2, whh2, T2, B2, H2, nd2, cuda)
    ref2 = _run(lib, False, gx2, whh2, T2, B2, H2, nd2, cuda)
    _same(got2, ref2)
    # and the first shape once more over the leftovers of the small one
    _same(_run(lib, True, gx, whh, T, B, H, ndir, cuda), ref)


def test_noalloc_entry_refuses_shapes_beyond_its_buffer(cuda):
    from video_classification_amd import _lib
    lib = _lib.lib()
    lib.vc_last_error.restype = ctypes.c_char_p
    # 3 batch chunks x 2 directions x 512 x 256 B = 768 KiB > 512 KiB; the shape itself is one the kernel supports
    assert lib.vc_lstm_persist_supported(1, 96, 512, 2) == 1
    rc = lib.vc_lstm_persist_fwd_noalloc(1, None, None, None, 1024, None, None, None, 4, 96, 512, 2, None)
    assert rc == -1
    assert b"exchange buffer" in lib.vc_last_error()
