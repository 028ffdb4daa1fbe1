import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running case")


def _gpu_available():
    try:
        from of_dis_amd import capi
        return capi.lib().ofdis_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """The HIP library on a box with a GPU.  No fallback: a missing library is a failure, not a skip."""
    from of_dis_amd import capi
    L = capi.lib()  # raises if libofdis_hip.so is missing
    if L.ofdis_device_count() < 1:
        pytest.skip("no HIP device visible")
    return capi


@pytest.fixture(scope="session")
def orc():
    """C restatement in the wave64 reduction order (the order the HIP kernels use)."""
    import oracle
    o = oracle.c_oracle()
    o.set_reduce_order(True)
    return o


@pytest.fixture(params=["single-wave", "multi-wave", "multi-wave-split", "cross-cu", "pipelined-strips"])
def tv_variant(gpu, request):
    """The mappings of the fused TV kernel (ofdis_fused.hip): one wavefront walking all fixed-point iterations of a
    frame group; a workgroup with one wavefront per iteration; the same with each iteration divided between a producer and
    a solver wavefront (up to 6 iterations); one workgroup of four wavefronts per iteration, the iterations of a group on
    different CUs with the du/dv rows handed over through global memory (what the launcher picks by itself for these
    test sizes); and the throughput form of the second one: a wavefront per iteration over STRIPS of two frames (what large
    batches run).  Bit-identical results are required of all (exact contract)."""
    small = request.param in ("multi-wave", "multi-wave-split", "cross-cu")
    old = gpu.set_tuning(fused_mw_max=(1 << 30) if small else 0,
                         fused_split=1 if request.param == "multi-wave-split" else 0,
                         fused_xcu_max=(1 << 30) if request.param == "cross-cu" else 0,
                         fused_tp_pipe=2 if request.param == "pipelined-strips" else 0,
                         fused_strip=2 if request.param == "pipelined-strips" else 0)
    yield request.param
    gpu.restore_tuning(old)
