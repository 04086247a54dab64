"""GPU parity of the FUSED multi-level step (nic_fused_ml_forward_backward / nic_fused_ml_forward, csrc/fused_q16.hpp::QML): several level pairs per
sample in ONE launch - BASELINE config 2's "16-level grid" as the extension it is (the reference reads one pair per sample, fp_def.py:24-34).

No reference semantics, so the pin is the composition of what IS pinned: the oracle's ``create_decoder_input`` for every pair (the reference's
encoding, image_compression.py:71-100, golden-checked in tests/test_oracle_golden.py), concatenated, into the oracle's decoder - the bf16-emulating one
(oracle/nic_oracle.py::mlp_forward_backward_bf16, the arithmetic of the plain-bf16 kernels) at 1e-3 of each tensor's largest magnitude (outputs 2e-3,
grid gradients 3e-3: see tests/test_gpu_bf16.py), AND the plain fp32 one at 3e-2 so that a wrong layout cannot hide behind the precision.  Every fused
instantiation, both positional encodings, tensor / in-kernel / no noise, unaligned crops that touch the far edge of every level, ragged extents; the
4K shape the bench times; the host loop (MultiLevelField) in fused mode."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nic_oracle as O  # noqa: E402  (checker only)
from tests.test_gpu_parity import relmax  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from neural_image_compression_v2_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _field(size, levels, C, seed):
    from neural_image_compression_v2_amd.multilevel import level_nodes
    g = torch.Generator().manual_seed(seed)
    fp = []
    for l in range(levels):
        for nodes in level_nodes(size, l):
            fp.append(torch.rand(C, nodes[1], nodes[0], generator=g) - 0.498)
    return fp


def _check(tag, out_y, out_loss, out_gfp, out_gmlp, ref, ref32, nl):
    y, loss, gfp, gmlp, _ = ref
    y32, loss32, gfp32, gmlp32, _ = ref32
    names = [f"{k}{i + 1}" for i in range(nl) for k in ("W", "b")]
    errs = {"y": relmax(out_y, y), "loss": relmax(out_loss, loss)}
    e32 = {"y": relmax(out_y, y32), "loss": relmax(out_loss, loss32)}
    for i, (a, b, c) in enumerate(zip(out_gfp, gfp, gfp32)):
        errs[f"g{i}"] = relmax(a, b)
        e32[f"g{i}"] = relmax(a, c)
    for nme, a, b, c in zip(names, out_gmlp, gmlp, gmlp32):
        errs[nme] = relmax(a, b)
        e32[nme] = relmax(a, c)
    print(f"\n[{tag}] vs emulating oracle: " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    print(f"[{tag}] vs fp32 oracle:      " + " ".join(f"{k}={v:.1e}" for k, v in e32.items()))
    bad = {k: v for k, v in errs.items() if not (np.isfinite(v) and v <= (2e-3 if k == "y" else (3e-3 if k.startswith("g") else 1e-3)))}
    assert not bad, f"{tag}: against the bf16-emulating oracle {bad}"
    bad32 = {k: v for k, v in e32.items() if not (np.isfinite(v) and v <= 3e-2)}
    assert not bad32, f"{tag}: against the fp32 oracle {bad32}"


ML_CASES = [
    # levels, C, NL, tri, image size, extent, origins, noise
    (2, 4, 3, True, (128, 96), (48, 40), [(3, 5), (80, 56)], "tensor"),
    (3, 4, 3, False, (256, 256), (64, 64), [(0, 0), (192, 192)], "kernel"),
    (5, 4, 3, True, (1024, 1024), (70, 33), [(1, 2), (954, 991), (512, 300)], "kernel"),
    (5, 4, 3, False, (1024, 1024), (64, 64), [(0, 0), (960, 960)], "none"),
    (2, 4, 5, True, (128, 96), (48, 40), [(3, 5), (80, 56)], "kernel"),
    (3, 4, 5, True, (256, 192), (37, 21), [(3, 5), (219, 171)], "tensor"),
    (2, 12, 3, False, (200, 136), (48, 40), [(3, 5), (152, 96)], "kernel"),
    (3, 12, 3, True, (256, 256), (64, 48), [(17, 101), (192, 208)], "kernel"),
    (5, 4, 3, True, (1024, 1024), (1, 1), [(1023, 1023)], "kernel"),                      # a single sample in the far corner of every level
]


@pytest.mark.parametrize("case", ML_CASES, ids=lambda c: f"L{c[0]}-C{c[1]}-NL{c[2]}-{'tri' if c[3] else 'sin'}-{c[5][0]}x{c[5][1]}-{c[7]}")
def test_fused_multilevel_step_matches_the_oracle_composition(dev, case):
    from neural_image_compression_v2_amd import _lib, fused
    L, C, NL, tri, size, ext, origins, noise_kind = case
    P = 6
    fp = _field(size, L, C, 11)
    cin = L * (5 * C + 2 * P) + 1
    mlp = O.init_mlp(cin, 64, torch.Generator().manual_seed(5), n_linear=NL)
    n = len(origins) * ext[0] * ext[1]
    g = torch.Generator().manual_seed(9)
    target = torch.rand(n, 3, generator=g)
    noise = None
    mode = {"none": _lib.NIC_NOISE_NONE, "tensor": _lib.NIC_NOISE_TENSOR, "kernel": _lib.NIC_NOISE_KERNEL}[noise_kind]
    if noise_kind == "tensor":
        noise = (torch.rand(n, cin, generator=g) - 0.5) / 256
    elif noise_kind == "kernel":
        noise = O.kernel_noise(n, cin, 8, 7, 3, layout=(2, 1, C, P, L))
    ref = O.multilevel_forward_backward(fp, mlp, origins, ext, target, noise, P, tri, emulate="bf16")
    ref32 = O.multilevel_forward_backward(fp, mlp, origins, ext, target, noise, P, tri)
    geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=ext, num_crops=len(origins), channels=C, pe_channels=P, use_tri_pe=tri,
                             noise_mode=mode, noise_seed=7, noise_offset=3)
    out = fused.fused_ml_forward_backward(geo, [t.to(dev) for t in fp], origins, [t.to(dev) for t in mlp.tensors()], target.to(dev),
                                          noise.to(dev) if noise_kind == "tensor" else None, want_y=True)
    _check(f"ml L{L} C{C} NL{NL}", out.y, out.loss, out.grad_fp, out.grad_mlp, ref, ref32, NL)
    # forward-only entry point on the same samples (no noise): against the emulating oracle's outputs
    yf = fused.fused_ml_forward(geo, [t.to(dev) for t in fp], origins, [t.to(dev) for t in mlp.tensors()])
    ref_f = O.multilevel_forward_backward(fp, mlp, origins, ext, target, None, P, tri, emulate="bf16")
    assert relmax(yf, ref_f[0]) <= 2e-3
    # gradients ADD into a caller's buffers, decoder gradients and loss are overwritten; run to run the decoder gradients are bit-stable
    out2 = fused.fused_ml_forward_backward(geo, [t.to(dev) for t in fp], origins, [t.to(dev) for t in mlp.tensors()], target.to(dev),
                                           noise.to(dev) if noise_kind == "tensor" else None, grads=[t.clone() for t in out.grad_fp])
    for a, b in zip(out2.grad_fp, out.grad_fp):
        assert relmax(a, 2 * b) < 1e-5
    for a, b in zip(out2.grad_mlp, out.grad_mlp):
        assert torch.equal(a, b)


def test_fused_multilevel_refuses_what_it_has_no_kernel_for(dev):
    from neural_image_compression_v2_amd import _lib, fused
    fp = [t.to(dev) for t in _field((256, 256), 4, 4, 1)]
    mlp = O.init_mlp(4 * 32 + 1, 64, torch.Generator().manual_seed(5), n_linear=3)
    geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(16, 16), num_crops=1, channels=4)
    with pytest.raises(_lib.Unsupported):
        fused.fused_ml_forward_backward(geo, fp, [[0, 0]], [t.to(dev) for t in mlp.tensors()], torch.zeros(256, 3, device=dev))
    assert fused.ml_is_fused(5, 4, 6, 64, 3) and not fused.ml_is_fused(5, 12, 6, 64, 3) and not fused.ml_is_fused(5, 4, 6, 64, 5) and not fused.ml_is_fused(2, 4, 6, 32, 3)


def test_fused_multilevel_4k_properties(dev):
    """the shape bench.py --workload multilevel times: 3840 x 2160, 5 pairs, C = 4, one launch.  Oracle-checked windows of the outputs, the loss against an
    independent reduction of the kernel's own outputs, a strip of the image as its own launch with the global sample numbering (same outputs bit for
    bit, same noise), and the grid gradients of that strip against the emulating oracle."""
    from neural_image_compression_v2_amd import _lib, fused
    H, W, L, C, P = 2160, 3840, 5, 4, 6
    fp = _field((H, W), L, C, 2)
    cin = L * (5 * C + 2 * P) + 1
    mlp = O.init_mlp(cin, 64, torch.Generator().manual_seed(5), n_linear=3)
    params = [t.to(dev) for t in mlp.tensors()]
    fpd = [t.to(dev) for t in fp]
    target = torch.rand(H * W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(H, W), num_crops=1, channels=C, pe_channels=P,
                             noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=7, noise_offset=1)
    out = fused.fused_ml_forward_backward(geo, fpd, [[0, 0]], params, target, want_y=True)
    torch.cuda.synchronize()
    loss_ind = float(((out.y.double() - target.double()) ** 2).mean())
    assert abs(float(out.loss) - loss_ind) < 1e-5 * loss_ind
    # windows: 24 x 32 samples at the four corners and the centre, against the emulating oracle (the in-kernel noise restated for the global sample ids)
    yk = out.y.reshape(H, W, 3)
    for (x0, y0) in [(0, 0), (H - 24, W - 32), (0, W - 32), (H - 24, 0), (1064, 1900)]:
        rows = (torch.arange(x0, x0 + 24)[:, None] * W + torch.arange(y0, y0 + 32)[None, :]).reshape(-1)
        # the generator is counter-based: noise of sample id s needs no other sample
        noise = torch.cat([O.kernel_noise(32, cin, 8, 7, 1, sample_base=int(r0), layout=(2, 1, C, P, L)) for r0 in rows.reshape(24, 32)[:, 0]])
        x = O.multilevel_decoder_input(fp, [[x0, y0]], (24, 32), L, P, True)
        yr, _, _, _ = O.mlp_forward_backward_bf16(x + noise, mlp, torch.zeros(24 * 32, 3), 24 * 32)
        assert relmax(yk[x0:x0 + 24, y0:y0 + 32].reshape(-1, 3), yr) <= 2e-3, (x0, y0)
    # a strip of 64 rows as its own launch: global sample numbering (sample_base) and the global mean -> same outputs, and its gradients against the oracle
    x0, rows_n = 1024, 64
    geo_s = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(rows_n, W), num_crops=1, channels=C, pe_channels=P,
                               noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=7, noise_offset=1, sample_base=x0 * W, loss_scale=1.0 / (3.0 * H * W))
    t_s = target.reshape(H, W, 3)[x0:x0 + rows_n].reshape(-1, 3)
    outs = fused.fused_ml_forward_backward(geo_s, fpd, [[x0, 0]], params, t_s, want_y=True)
    assert torch.equal(outs.y, yk[x0:x0 + rows_n].reshape(-1, 3))
    noise = O.kernel_noise(rows_n * W, cin, 8, 7, 1, sample_base=x0 * W, layout=(2, 1, C, P, L))
    ref = O.multilevel_forward_backward(fp, mlp, [[x0, 0]], (rows_n, W), t_s.cpu(), noise, P, True, mean_over=H * W, emulate="bf16")
    ref32 = O.multilevel_forward_backward(fp, mlp, [[x0, 0]], (rows_n, W), t_s.cpu(), noise, P, True, mean_over=H * W)
    _check("ml 4K strip", outs.y, outs.loss, outs.grad_fp, outs.grad_mlp, ref, ref32, 3)
    # nothing outside the strip's cells moved: node rows of pair 0 further than one G1 cell (8 px) from the strip are zero
    g0 = outs.grad_fp[0]
    assert float(g0[:, :, : x0 // 4 - 2].abs().max()) == 0.0 and float(g0[:, :, (x0 + rows_n) // 4 + 3:].abs().max()) == 0.0


def test_multilevel_field_fused_fit(dev):
    """MultiLevelField in fused mode (the default where a kernel exists): a 60-step whole-image fit of a 256 x 192 image with 3 pairs of 4 channels, chunked
    passes included (gradients accumulate in place over the chunks); the loss falls like the layer-wise fp32 route's on the same schedule, the fused
    decode agrees with the fp32 decode to bf16 precision, nothing leaves the quantiser's range"""
    from neural_image_compression_v2_amd.multilevel import MultiLevelField
    S = (256, 192)
    u, v = torch.linspace(0, 1, S[0]), torch.linspace(0, 1, S[1])
    img = torch.stack([0.5 + 0.25 * torch.sin(6.28 * (c + 1) * u)[:, None] * torch.cos(6.28 * (c + 2) * v)[None, :] for c in range(3)]).clamp(0, 1)
    tgt = img.permute(1, 2, 0).to(dev)
    finals = {}
    for mode in ("fused", "layerwise"):
        f = MultiLevelField(S, 3, channels=4, hidden=64, n_linear=3, device=dev, seed=0, fused_step=mode == "fused")
        assert f.fused_step == (mode == "fused")
        f.set_schedule(60)
        losses = []
        for step in range(60):
            if step % 2 == 1:
                losses.append(float(f.train_step([[0, 0]], S, tgt.reshape(-1, 3))))
            else:
                tot = 0.0
                for k, x0 in enumerate((0, 128)):
                    tot += float(f.train_step([[x0, 0]], (128, 192), tgt[x0:x0 + 128].reshape(-1, 3), accumulate=k > 0, scale=0.5, step=k == 1))
                losses.append(tot)
        assert losses[-1] < 0.3 * losses[0], (mode, losses)
        lo = -(2 ** 8 - 1) / 2 ** 9
        assert all(float(g.detach().min()) >= lo and float(g.detach().max()) <= 0.5 for g in f.fp)
        finals[mode] = (losses, f)
    lf, ll = finals["fused"][0], finals["layerwise"][0]
    assert abs(lf[-1] - ll[-1]) < 0.15 * ll[-1], (lf[-5:], ll[-5:])          # different noise streams (in-kernel vs torch.rand_like), same fit
    f = finals["fused"][1]
    a, b = f.decode(tile=100, fused_forward=True), f.decode(tile=100)
    assert relmax(a, b) < 5e-3
