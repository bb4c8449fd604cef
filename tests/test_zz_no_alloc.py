"""dfx_resid_update_noalloc (include/dfx.h): the entry point for callers that cannot allocate.  Its scratch is a static
__device__ array inside libdfx.so; it must give, bit for bit, what dfx_resid_update gives with a caller workspace - on
two calls in a row, the second on a smaller input (the scratch keeps the first call's contents)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both_ways(dims, seed):
    from datafusion_amd import _cabi as cabi
    from datafusion_amd.kernels import HipKernels
    n_t, n_l, n_z, X, Y, r = dims
    torch.manual_seed(seed)
    ref = HipKernels(n_t, n_l, n_z, X, Y, r, DEV)
    new = HipKernels(n_t, n_l, n_z, X, Y, r, DEV)
    ref.new_factored_residual()
    new.new_factored_residual()
    new.zab = None                                   # the no-alloc entry point brings its own
    A = torch.randn(n_z, X, Y, device=DEV)
    B = torch.randn(n_t, n_l, X // r, Y // r, device=DEV)
    A2, B2 = torch.randn_like(A), torch.randn_like(B)
    for k in (ref, new):
        k.scal[cabi.S_ALPHA] = 0.37
    slots = [cabi.S_RR, cabi.S_RA2, cabi.S_RB2, cabi.S_RAB]
    # first residual (accumulate = 0, host coefficient), then an update (accumulate = 1, coefficient from the device)
    ref.resid_update(A, B, 0.7, -1.3, 2.0, None, False)
    new.resid_update_noalloc(A, B, 0.7, -1.3, 2.0, None, False)
    assert torch.equal(ref.ra, new.ra) and torch.equal(ref.rb, new.rb)
    assert torch.equal(ref.scal[slots], new.scal[slots]) and float(new.scal[cabi.S_RR]) > 0
    ref.resid_update(A2, B2, 0.4, 0.9, -1.0, cabi.S_ALPHA, True)
    new.resid_update_noalloc(A2, B2, 0.4, 0.9, -1.0, cabi.S_ALPHA, True)
    assert torch.equal(ref.ra, new.ra) and torch.equal(ref.rb, new.rb)
    assert torch.equal(ref.scal[slots], new.scal[slots])
    # and against the definition: ||bcast(ra) + bcast(rb)||^2 in float64
    r5 = new.ra.double()[None, None] + new.rb.double().repeat_interleave(r, 2).repeat_interleave(r, 3)[:, :, None]
    rr = float((r5 ** 2).sum())
    assert abs(float(new.scal[cabi.S_RR]) - rr) <= 1e-9 * rr


def test_noalloc_resid_update_matches_workspace_variant_twice_second_smaller():
    _both_ways((6, 4, 5, 64, 128, 4), seed=0)        # Xs*Ys = 512, 24 slabs
    _both_ways((3, 2, 3, 24, 20, 4), seed=1)         # smaller: Xs*Ys = 30; the scratch still holds the first call's data
    _both_ways((5, 3, 2, 16, 24, 2), seed=2)         # r = 2


def test_noalloc_entry_point_under_hip_graph_capture():
    """The callers it exists for: after ONE eager call on the device (which resolves the scratch's address with
    hipGetSymbolAddress - not a stream operation, so not inside a capture), the entry point is captured into a HIP
    graph and replayed; each replay gives the bits of the workspace variant run eagerly."""
    from datafusion_amd import _cabi as cabi
    from datafusion_amd.kernels import HipKernels
    dims = (6, 4, 5, 64, 128, 4)
    torch.manual_seed(3)
    ref, new = HipKernels(*dims, DEV), HipKernels(*dims, DEV)
    for k in (ref, new):
        k.new_factored_residual()
        k.scal[cabi.S_ALPHA] = 0.25
    A = torch.randn(5, 64, 128, device=DEV)
    B = torch.randn(6, 4, 16, 32, device=DEV)
    ref.resid_update(A, B, 0.7, -1.3, 2.0, None, False)
    new.resid_update_noalloc(A, B, 0.7, -1.3, 2.0, None, False)          # eager warm-up: resolves the symbol
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        new.resid_update_noalloc(A, B, 0.4, 0.9, 0.0, cabi.S_ALPHA, True)     # coefficient 0: state unchanged
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        new.resid_update_noalloc(A, B, 0.4, 0.9, -1.0, cabi.S_ALPHA, True)    # recorded, not executed
    slots = [cabi.S_RR, cabi.S_RA2, cabi.S_RB2, cabi.S_RAB]
    for _ in range(3):
        g.replay()
        ref.resid_update(A, B, 0.4, 0.9, -1.0, cabi.S_ALPHA, True)
        torch.cuda.synchronize()
        assert torch.equal(ref.ra, new.ra) and torch.equal(ref.rb, new.rb)
        assert torch.equal(ref.scal[slots], new.scal[slots])


def test_noalloc_rejects_sizes_beyond_its_scratch():
    from datafusion_amd import _cabi as cabi
    lib = cabi.lib()
    g = cabi.Geom(2, 2, 1, 1028, 1028, 4, 0)          # 257 * 257 = 66049 blocks > DFX_NOALLOC_MAX_PS
    assert (1028 // 4) ** 2 > cabi.NOALLOC_MAX_PS
    dummy = torch.zeros(16, device=DEV)
    scal = torch.zeros(cabi.NSCAL, dtype=torch.float64, device=DEV)
    rc = lib.dfx_resid_update_noalloc(C.byref(g), dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(),
                                      1.0, 1.0, 1.0, None, 0, scal.data_ptr(),
                                      torch.cuda.current_stream(torch.device(DEV)).cuda_stream)
    assert rc == -1 and b"DFX_NOALLOC_MAX_PS" in lib.dfx_last_error()
    torch.cuda.synchronize()
