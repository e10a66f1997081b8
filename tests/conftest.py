import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One fgpu context per test session; fails loudly when no HIP device is present."""
    from falkordb_amd.engine import Context
    c = Context(0)
    yield c
    c.close()


class _BenchGraphs:
    """The R-MAT graphs bench.py times (seed 0x5EED1234 + scale, edge factor 16), built once per test session and kept
    resident with their transpose and a host copy for the oracle: RMAT-26 alone is 20 s of generation + export, and the
    parity tests at the BASELINE sizes (tests/test_gpu_scale.py, tests/test_gpu_tiled.py) all start from these."""

    def __init__(self, ctx):
        self.ctx, self._g = ctx, {}

    def __call__(self, scale):
        if scale not in self._g:
            import oracle
            A = self.ctx.mat_rmat(scale, 16, 0x5EED1234 + scale)
            At = A.transpose()
            rp, ci, _ = A.export_csr()
            self._g[scale] = (A, At, oracle.CSR(A.nrows, A.ncols, rp, ci))
        return self._g[scale]

    def khop_layers(self, scale):
        """bench.py khop_inputs' dirty layers for this graph: (dp, dm) hypersparse on the device + their host CSRs."""
        key = ("layers", scale)
        if key not in self._g:
            import numpy as np
            import oracle
            A, _, _ = self(scale)
            n = A.nrows
            dm0 = A.sample(0xD3170 + scale, 1000)
            rng = np.random.default_rng(0xADD5 + scale)
            k = max(1, A.nvals // 1000)
            raw = self.ctx.mat_from_coo(n, n, rng.integers(0, n, k, dtype=np.uint64), rng.integers(0, n, k, dtype=np.uint64))
            dp0 = raw.merge(None, A)
            raw.free()
            host = []
            dev = []
            for m in (dp0, dm0):
                rp, ci, _ = m.export_csr()
                host.append(oracle.CSR(n, n, rp, ci))
                deg = np.diff(rp.astype(np.int64))
                rows = np.nonzero(deg)[0].astype(np.uint64)
                short = np.concatenate([[0], np.cumsum(deg[deg > 0])]).astype(np.uint64)
                dev.append(self.ctx.mat_from_csr(n, n, short, ci, hyper_rows=rows))
                m.free()
            self._g[key] = (dev[0], dev[1], host[0], host[1])
        return self._g[key]


@pytest.fixture(scope="session")
def bench_graphs(ctx):
    return _BenchGraphs(ctx)
