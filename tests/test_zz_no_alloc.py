"""GPU: the entry point that allocates nothing (icp_amd_centroid_static, include/icp_amd.h) against the existing entry point
it is a variant of (icp_amd_centroid), through the C ABI: bit-identical results on two calls in a row, the second with a
smaller input (the calls share one static device scratch array, which keeps the first call's contents), on the null stream
and on a stream of the caller's, and a clean error beyond the stated maximum."""
import numpy as np
import pytest

import icp_amd
from icp_amd import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _static_centroid(A, stream=None):
    dev = torch.device("cuda", 0)
    a = torch.from_numpy(np.ascontiguousarray(A)).to(dev)
    out = torch.full((3,), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    if stream is None:
        icp_amd.centroid_static(a.data_ptr(), len(A), out.data_ptr(), device=0, stream=None)
        torch.cuda.synchronize()
    else:
        icp_amd.centroid_static(a.data_ptr(), len(A), out.data_ptr(), device=0, stream=stream.cuda_stream)
        stream.synchronize()
    return out.cpu().numpy()


def test_static_centroid_equals_allocating_entry_point_two_calls_in_a_row():
    P, _ = synth.make_pair(300000, 1000, seed=91)
    big, small = P * 3.0 + 7.0, P[:4321] - 11.0
    want_big, want_small = icp_amd.centroid(big), icp_amd.centroid(small)
    got_big = _static_centroid(big)                 # fills 293 chunk partials of the static scratch
    got_small = _static_centroid(small)             # 5 chunks: the rest of the scratch still holds the first call's sums
    assert np.array_equal(got_big, want_big)
    assert np.array_equal(got_small, want_small)
    # once more on a stream of the caller's, sizes that are not multiples of the 1024-row chunk, down to one row
    st = torch.cuda.Stream(torch.device("cuda", 0))
    for rows in (P[:1], P[:1023], P[:1025], P[:100000], P[:77]):
        assert np.array_equal(_static_centroid(rows, st), icp_amd.centroid(rows)), len(rows)


def test_static_centroid_allocates_nothing_and_runs_under_stream_capture(monkeypatch):
    """The property the entry point exists for.  (a) Free device memory (hipMemGetInfo) and the bytes the library's own block
    cache holds are the same before and after a call, with the cache switched off (ICP_AMD_CACHE_MB=0) so that nothing could
    hide an allocation.  (b) The call is legal inside hipStreamBeginCapture ... EndCapture -- where hipMalloc is not -- after
    icp_amd_init(device) (which takes the scratch address with hipGetSymbolAddress up front, as the header asks of capture
    users): captured into a graph, replayed on new input in the same buffers, same bits as icp_amd_centroid each time."""
    monkeypatch.setenv("ICP_AMD_CACHE_MB", "0")
    dev = torch.device("cuda", 0)
    icp_amd.init(0)
    icp_amd.cache_trim()
    P, _ = synth.make_pair(200000, 1000, seed=92)
    a = torch.from_numpy(np.ascontiguousarray(P)).to(dev)
    out = torch.zeros(3, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream(dev)
    icp_amd.centroid_static(a.data_ptr(), len(P), out.data_ptr(), device=0, stream=st.cuda_stream)   # code objects loaded
    st.synchronize()
    torch.cuda.synchronize()
    free0, held0 = torch.cuda.mem_get_info(dev)[0], icp_amd.cache_trim()
    for rows in (len(P), 4097, len(P)):
        icp_amd.centroid_static(a.data_ptr(), rows, out.data_ptr(), device=0, stream=st.cuda_stream)
    st.synchronize()
    free1, held1 = torch.cuda.mem_get_info(dev)[0], icp_amd.cache_trim()
    assert free1 == free0 and held0 == held1 == 0, (free0, free1, held0, held1)
    assert np.array_equal(out.cpu().numpy(), icp_amd.centroid(P))
    # (b) under stream capture
    out.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        icp_amd.centroid_static(a.data_ptr(), len(P), out.data_ptr(), device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), np.zeros(3))        # captured, not run
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), icp_amd.centroid(P))
    P2 = P[::-1] * 2.5 - 1.0
    a.copy_(torch.from_numpy(np.ascontiguousarray(P2)))
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), icp_amd.centroid(P2))


def test_static_centroid_rejects_what_its_scratch_cannot_hold():
    limit = icp_amd.centroid_static_max_rows()
    assert limit == 16 << 20
    dev = torch.device("cuda", 0)
    a = torch.zeros((8, 3), dtype=torch.float64, device=dev)
    out = torch.zeros(3, dtype=torch.float64, device=dev)
    with pytest.raises(icp_amd.IcpAmdError, match="exceed the static scratch"):
        icp_amd.centroid_static(a.data_ptr(), limit + 1, out.data_ptr())      # refused before anything is launched
    with pytest.raises(icp_amd.IcpAmdError):
        icp_amd.centroid_static(a.data_ptr(), 0, out.data_ptr())
    with pytest.raises(icp_amd.IcpAmdError):
        icp_amd.centroid_static(0, 8, out.data_ptr())
