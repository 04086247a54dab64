"""GPU parity of the PLAIN-bf16 product mode (NIC_FLAG_BF16, csrc/fused_q16.hpp: 8 waves x 16 samples, every layout, 3 or 5 Linear
layers) against the precision-emulating oracle (oracle/nic_oracle.py::mlp_forward_backward_bf16: the same arithmetic with a rounding
to bf16 at every point the kernel rounds - SURVEY 7 step 2), all through the C ABI.

Tolerances.  Against the emulating oracle: 1e-3 of the tensor's largest magnitude (the north star's figure).  What separates the two is
(a) fp32 summation order, ~1e-6, and (b) a bf16 rounding that falls the other way because GELU / sigmoid differ in the last fp32 bits -
one operand off by 2^-9 relative; over a reduction of many samples that stays far below 1e-3, on a single output it can reach ~1e-3 of
that OUTPUT, hence outputs are held element-wise at 2e-3 absolute (sigmoid range 1).  Against the fp32 oracle (no emulation) the mode sits
where bf16 arithmetic puts it: ~1e-2; bounded here at 3e-2 so that a wrong layout cannot hide behind the precision."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nic_oracle as O  # noqa: E402  (checker only)
from tests.test_gpu_parity import _pyramid, relmax  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from neural_image_compression_v2_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def check_step(out, ref, ref32, nl, tag, tol=1e-3, tol32=3e-2):
    names = [f"{k}{i + 1}" for i in range(nl) for k in ("W", "b")]
    errs = {"y": relmax(out.y, ref.y), "loss": relmax(out.loss, ref.loss), "g0": relmax(out.grad_g0, ref.grad_g0), "g1": relmax(out.grad_g1, ref.grad_g1)}
    for nme, a, b in zip(names, out.grad_mlp, ref.grad_mlp):
        errs[nme] = relmax(a, b)
    e32 = {"y": relmax(out.y, ref32.y), "loss": relmax(out.loss, ref32.loss), "g0": relmax(out.grad_g0, ref32.grad_g0), "g1": relmax(out.grad_g1, ref32.grad_g1)}
    for nme, a, b in zip(names, out.grad_mlp, ref32.grad_mlp):
        e32[nme] = relmax(a, b)
    print(f"\n[{tag}] vs emulating oracle: " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    print(f"[{tag}] vs fp32 oracle:      " + " ".join(f"{k}={v:.1e}" for k, v in e32.items()))
    # grid gradients: a node sums few samples (four per G0 node at mip 1, one at mip >= 2), so ONE bf16 rounding of a dZ that falls the other
    # way (the kernel's GELU and torch's differ in the last fp32 bits) shows at up to 2^-8 of that sample's contribution: 3e-3
    bad = {k: v for k, v in errs.items() if not (np.isfinite(v) and v <= (2e-3 if k == "y" else (3e-3 if k in ("g0", "g1") else tol)))}
    assert not bad, f"{tag}: against the bf16-emulating oracle {bad}"
    bad32 = {k: v for k, v in e32.items() if not (np.isfinite(v) and v <= tol32)}
    assert not bad32, f"{tag}: against the fp32 oracle {bad32}"


BF16_CASES = [
    # dim, method, tri, base | (base, fl, mip), extent, origins, noise, passes
    (2, 1, True, 64, (64, 64), [(17, 101), (0, 0), (192, 192)], "tensor", 1),
    (2, 1, True, 64, (37, 21), [(3, 5), (200, 100)], "kernel", 1),
    (2, 1, False, 64, (40, 24), [(3, 5), (20, 0)], "none", 1),
    (2, 1, True, 64, (256, 256), [(0, 0), (0, 0)], "kernel", 1),          # the reference's default crop shape
    (2, 1, False, 64, (64, 64), [(0, 0), (64, 128)], "kernel", 2),         # aligned origins, two passes
    (2, 1, True, 64, (1, 1), [(5, 250)], "kernel", 1),                    # a single sample
    (2, 1, True, (64, 0, 1), (40, 24), [(3, 5), (50, 30)], "kernel", 1),  # mip pyramid levels: step 1/2, 1, 2 (unweighted G1, Q6), 4
    (2, 1, True, (64, 0, 2), (20, 24), [(3, 5), (20, 7)], "tensor", 1),
    (2, 1, False, (64, 0, 3), (10, 9), [(3, 5), (12, 0)], "kernel", 1),
    (2, 1, True, (64, 1, 6), (2, 3), [(0, 1)], "kernel", 1),
    (3, 3, True, 16, (8, 8, 8), [(3, 5, 9), (56, 0, 31)], "tensor", 1),
    (3, 3, True, 16, (5, 3, 7), [(0, 0, 0), (59, 61, 57)], "kernel", 1),
    (3, 3, True, 16, (32, 32, 32), [(3, 5, 9), (8, 0, 31)], "kernel", 1),  # the reference's sweep crop (packed tiling)
    (3, 3, True, 16, (16, 12, 8), [(0, 0, 0), (16, 4, 8)], "none", 2),
    (3, 4, False, 16, (8, 8, 8), [(3, 5, 9), (56, 0, 31)], "kernel", 1),
    (3, 4, False, 16, (6, 5, 3), [(1, 2, 3)], "none", 1),
    (3, 4, False, 16, (32, 32, 32), [(1, 2, 3), (30, 11, 7)], "tensor", 1),
    (3, 4, False, 16, (64, 8, 4), [(0, 0, 0)], "kernel", 3),
    (3, 3, True, (16, 0, 1), (6, 5, 7), [(1, 2, 3), (20, 9, 0)], "kernel", 1),
    (3, 4, False, (16, 0, 2), (5, 4, 3), [(1, 2, 3)], "tensor", 1),
]


def _setup(case, nl, seed=77):
    dim, method, tri, base, extent, origins, noise_kind, passes = case
    fl, mip = 0, 0
    if isinstance(base, tuple):
        base, fl, mip = base
    fp, _ = _pyramid(dim, base, 12, seed=9, no_mip=(mip == 0))
    g0, g1 = fp[2 * fl], fp[2 * fl + 1]
    step = O.step_number_of(mip, fl)
    cin = O.decoder_input_channels(12, 6, dim, method)
    g = torch.Generator().manual_seed(seed)
    mlp = O.init_mlp(cin, 64, generator=g, n_linear=nl)
    org_list = [o for o in origins for _ in range(passes)]              # the oracle lists a crop once per pass
    n = len(org_list) * int(np.prod(extent))
    target = torch.rand(n, 3, generator=g)
    return dim, method, tri, g0, g1, step, mip, cin, mlp, org_list, n, target, g


@pytest.mark.parametrize("nl", [3, 5])
@pytest.mark.parametrize("case", BF16_CASES, ids=lambda c: f"d{c[0]}m{c[1]}{'t' if c[2] else 's'}-{c[3]}-{'x'.join(map(str, c[4]))}-{c[6]}-p{c[7]}".replace(" ", ""))
def test_plain_bf16_step_matches_emulating_oracle(dev, case, nl):
    from neural_image_compression_v2_amd import _lib, fused
    extent, origins, noise_kind, passes = case[4], case[5], case[6], case[7]
    dim, method, tri, g0, g1, step, mip, cin, mlp, org_list, n, target, g = _setup(case, nl)
    noise, kw = None, {}
    if noise_kind == "tensor":
        noise = (torch.rand(n, cin, generator=g) - 0.5) / 256
        kw = dict(noise_mode=_lib.NIC_NOISE_TENSOR)
    elif noise_kind == "kernel":
        noise = O.kernel_noise(n, cin, 8, seed=0x1234567890AB, offset=42, sample_base=1000, quarter=True)
        kw = dict(noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=0x1234567890AB, noise_offset=42, sample_base=1000)
    ref = O.forward_backward(g0, g1, mlp, org_list, extent, step, mip, target, noise, 6, method=method, use_tri_pe=tri, emulate="bf16")
    ref32 = O.forward_backward(g0, g1, mlp, org_list, extent, step, mip, target, noise, 6, method=method, use_tri_pe=tri)
    geo = fused.PathGeometry(dim=dim, method=method, step_number=step, mip_level=mip, extent=extent, num_crops=len(origins), use_tri_pe=tri,
                             bf16=True, passes=passes, **kw)
    params = [q.to(dev) for q in mlp.tensors()]
    nd = noise.to(dev) if noise_kind == "tensor" else None
    out = fused.fused_forward_backward(geo, g0.to(dev), g1.to(dev), origins, params, target.to(dev), nd, want_y=True)
    check_step(out, ref, ref32, nl, f"nl{nl} d{dim}m{method} {extent} {noise_kind}")
    out2 = fused.fused_forward_backward(geo, g0.to(dev), g1.to(dev), origins, params, target.to(dev), nd)
    assert relmax(out2.loss, out.loss) <= 1e-6
    for a, b in zip(out.grad_mlp, out2.grad_mlp):
        assert torch.equal(a, b), "decoder gradients are bit-stable run to run"


@pytest.mark.parametrize("nl", [3, 5])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("dm", [(2, 1), (3, 3), (3, 4)], ids=["2d", "m3", "m4"])
def test_plain_bf16_with_16_bit_grid_storage(dev, dm, dt, nl):
    """16-bit grid STORAGE under the plain-bf16 products, every layout (the reference's FP_NUM_DTYPE = 16, utils.py:301-313): the oracle
    works on the widened grids"""
    from neural_image_compression_v2_amd import _lib, fused
    dim, method = dm
    extent = (40, 24) if dim == 2 else (12, 7, 9)
    origins = [(3, 5), (100, 60)] if dim == 2 else [(3, 5, 9), (20, 0, 31)]
    case = (dim, method, True, 64 if dim == 2 else 16, extent, origins, "kernel", 1)
    dim, method, tri, g0, g1, step, mip, cin, mlp, org_list, n, target, g = _setup(case, nl, seed=5)
    g0s, g1s = g0.to(dt), g1.to(dt)
    noise = O.kernel_noise(n, cin, 8, seed=99, offset=3, quarter=True)
    tri = method != 4
    ref = O.forward_backward(g0s.float(), g1s.float(), mlp, org_list, extent, step, mip, target, noise, 6, method=method, use_tri_pe=tri, emulate="bf16")
    ref32 = O.forward_backward(g0s.float(), g1s.float(), mlp, org_list, extent, step, mip, target, noise, 6, method=method, use_tri_pe=tri)
    geo = fused.PathGeometry(dim=dim, method=method, step_number=step, mip_level=mip, extent=extent, num_crops=len(origins), use_tri_pe=tri,
                             bf16=True, noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=99, noise_offset=3)
    out = fused.fused_forward_backward(geo, g0s.to(dev), g1s.to(dev), origins, [q.to(dev) for q in mlp.tensors()], target.to(dev), want_y=True)
    assert out.grad_g0.dtype == torch.float32 and out.grad_g0.shape == g0.shape
    check_step(out, ref, ref32, nl, f"{dt} grids nl{nl} d{dim}m{method}")


def test_plain_bf16_image_targets_and_dy(dev):
    """the resident-image target formats (fp32 planar, uint8 planar, RGBX) and the dY entry point on the plain-bf16 kernels"""
    from neural_image_compression_v2_amd import _lib, fused
    g = torch.Generator().manual_seed(3)
    for dim, method in ((2, 1), (3, 4)):
        size = (96, 80) if dim == 2 else (24, 20, 16)
        extent = (40, 24) if dim == 2 else (10, 7, 5)
        origins = [(3, 5), (50, 30)] if dim == 2 else [(3, 5, 9), (12, 0, 2)]
        fp, _ = _pyramid(dim, 32 if dim == 2 else 8, 12, seed=4, no_mip=True)
        g0, g1 = fp[0], fp[1]
        cin = O.decoder_input_channels(12, 6, dim, method)
        mlp = O.init_mlp(cin, 64, generator=g)
        img8 = torch.randint(0, 256, (3, *size), generator=g, dtype=torch.uint8)
        den = 255.0 if dim == 2 else 256.0
        imgf = img8.float() / den
        sl = [tuple(slice(o[a], o[a] + extent[a]) for a in range(dim)) for o in origins]
        target = torch.cat([imgf[(slice(None), *s)].reshape(3, -1).T for s in sl])
        tri = method != 4
        ref = O.forward_backward(g0, g1, mlp, origins, extent, 0.25, 0, target, None, 6, method=method, use_tri_pe=tri, emulate="bf16")
        geo = fused.PathGeometry(dim=dim, method=method, step_number=0.25, mip_level=0, extent=extent, num_crops=len(origins), use_tri_pe=tri, bf16=True)
        params = [q.to(dev) for q in mlp.tensors()]
        base = fused.fused_forward_backward(geo, g0.to(dev), g1.to(dev), origins, params, target.to(dev), want_y=True)
        assert relmax(base.loss, ref.loss) <= 1e-3
        rgbx = (img8[0].int() | (img8[1].int() << 8) | (img8[2].int() << 16)).to(dev)
        for tgt in (fused.TargetImage(imgf.to(dev)), fused.TargetImage(img8.to(dev), den=den), fused.TargetImage(rgbx, den=den, rgbx=True)):
            out = fused.fused_forward_backward(geo, g0.to(dev), g1.to(dev), origins, params, tgt, want_y=True)
            assert torch.equal(out.y, base.y) and relmax(out.loss, base.loss) <= 1e-6
            for a, b in zip(out.grad_mlp, base.grad_mlp):
                assert torch.equal(a, b)
        # dY entry point: dy = dLoss/dy of the MSE reproduces the step's gradients
        n = target.shape[0]
        dy = (2.0 / (3 * n)) * (base.y - target.to(dev))
        y = fused.fused_grid_mlp(geo, g0.to(dev).requires_grad_(True), g1.to(dev).requires_grad_(True), origins, [p_.requires_grad_(True) for p_ in params])
        # (the forward of the autograd function runs the inference kernel - split / fp32 products; its backward the plain-bf16 kernel)
        grads = torch.autograd.grad(y, params, dy)
        for a, b in zip(grads, base.grad_mlp):
            assert relmax(a, b) <= 1e-5, "dY entry point"
