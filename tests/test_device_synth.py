"""Fixture tooling on the device (tools/synth_fill_hip.hip): the synthetic-weight filler and the gf4
quantiser running where the weights live.  They must produce exactly what their host counterparts produce -- the parity tests
on layer-reduced models use the host filler, the full-size Mixtral / DBRX runs the device one, and both claim the same model."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

from calm_amd import calmfile as cf
from calm_amd.host import HipBackend, HostModel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,dtype", [("tinyllama-1.1b", "fp16"), ("mistral-7b", "fp8"), ("llama-3-8b", "gf4"), ("mixtral-8x7b", "fp8")])
def test_device_filler_writes_the_host_fillers_bytes(hiplib, name, dtype):
    spec = cf.SPECS[name]
    L, seed = 1, 9
    host = dict(cf.synth_stream_big(spec, dtype, seed, L, reuse=False))
    model = HostModel(cf.stub_tensors(spec, dtype, L), dataclasses.replace(spec, n_layers=L).metadata(dtype))
    b = HipBackend(model, device_synth=(spec, dtype, seed, L))
    try:
        assert set(b._dev) == {n for n in host if n.startswith("model.")}
        for n, ptr in b._dev.items():
            want = np.ascontiguousarray(host[n]).view(np.uint8).reshape(-1)
            got = np.empty_like(want)
            hiplib.download_hip(got.ctypes.data, ptr, got.nbytes)
            assert np.array_equal(got, want), n
        # and the model decodes like the uploaded one
        up = HipBackend(HostModel(host, model.metadata))
        try:
            for pos, tok in enumerate([5, 77, 1234]):
                assert np.array_equal(b.forward(tok, pos, 0), up.forward(tok, pos, 0))
        finally:
            up.close()
    finally:
        b.close()


def test_device_filler_ragged_sizes(hiplib):
    """element counts that are not multiples of 4, of the 1 Mi-element chunk, or larger than one chunk"""
    lib = cf._dev_synth_lib()
    for kind, dtype, store in ((0, "fp8", np.uint8), (1, "fp16", np.uint16), (2, "gf4", np.uint32)):
        lut = cf._gf4_scale_lut(0.02) if dtype == "gf4" else cf._code_lut(dtype, 0.02)
        dlut = hiplib.upload_hip(lut.ctypes.data, lut.nbytes)
        for n in (1, 3, 4, 5, 1023, (1 << 20) - 1, (1 << 20) + 6, 3 * (1 << 20) + 2):
            want = np.zeros(n, dtype=store)
            cf._fill_codes(want, dtype, 0.02, 4242 + n)
            d = hiplib.alloc_hip(want.nbytes)
            lib.synth_fill_hip(d, n, kind, dlut, 4242 + n)
            got = np.empty_like(want)
            hiplib.download_hip(got.ctypes.data, d, got.nbytes)
            hiplib.free_hip(d)
            assert np.array_equal(got, want), (dtype, n)
        hiplib.free_hip(dlut)


def test_device_gf4_quantiser_is_calmfiles(hiplib):
    """quantize_gf4_hip == calmfile.quantize_gf4 (== the reference converter's gf4(), tests/test_calmfile.py) word for word"""
    lib = cf._dev_synth_lib()
    rng = np.random.default_rng(3)
    parts = [rng.standard_normal((4096, 64)).astype(np.float32) * s for s in (0.02, 1.0, 300.0, 1e-6)]
    edge = np.zeros((8, 64), dtype=np.float32)
    edge[1, :8] = [1, -1, 0.5, -0.5, 0.25, -0.25, 0.874, -0.876]  # the clamp at +7 and codes around rounding ties
    edge[2, :8] = [-2, 2, 1, -1, 0.24, 0.26, 1.74, 1.76]
    edge[3, :8] = 1e-30  # scale rounds to zero: every code 4 (0 / 0 -> nan -> 0)
    edge[4, :8] = [70000, 1, 2, 3, 4, 5, 6, 7]  # scale beyond e5m2's finite range -> inf -> codes 4
    w = np.concatenate(parts + [edge])
    want = cf.quantize_gf4(w).view(np.uint32).reshape(-1)
    din = hiplib.upload_hip(w.ctypes.data, w.nbytes)
    dout = hiplib.alloc_hip(want.nbytes)
    lib.quantize_gf4_hip(din, dout, want.size)
    got = np.empty_like(want)
    hiplib.download_hip(got.ctypes.data, dout, got.nbytes)
    hiplib.free_hip(din), hiplib.free_hip(dout)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:5], [hex(x) for x in got[bad[:5]]], [hex(x) for x in want[bad[:5]]])
