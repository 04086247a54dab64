"""GPU parity tests of the layer-wise general route (csrc/decoder_general.hip, nic_decoder_general_*): ColorDecoder of any Cin /
HIDDEN_LAYER_CHANNELS / depth on explicit inputs against the CPU oracle (fp64 autograd of oracle.mlp_forward), and the host loop
(ImageCompression) on flag combinations the fused kernels do not specialise - var2.py:68-72 - where it must fall back to
nic_encode + the general decoder + nic_encode_backward instead of refusing.

Tolerances: fp32 fmaf chains against fp64: 2e-5 of each tensor's largest magnitude (measured ~1e-6); the north star allows 1e-3.
"""
import ctypes
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nic_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from neural_image_compression_v2_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def relmax(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def oracle_fwd_bwd(x, mlp, dy):
    """fp64 autograd of the oracle's decoder: y, dx, parameter gradients for the upstream gradient dy"""
    xd = x.double().requires_grad_(True)
    p = O.MLPParams([w.double().requires_grad_(True) for w in mlp.w], [b.double().requires_grad_(True) for b in mlp.b])
    y = O.mlp_forward(xd, p)
    (y * dy.double()).sum().backward()
    return y.detach(), xd.grad, [t.grad for t in p.tensors()]


CASES = [
    # cin, hidden, n_linear, n
    (73, 64, 3, 1000),        # the reference's shape through the general kernels
    (40, 32, 3, 777),         # HIDDEN_LAYER_CHANNELS = 32
    (73, 128, 3, 5000),       # HIDDEN_LAYER_CHANNELS = 128
    (127, 64, 5, 2049),       # method 3's Cin, "4 x 64"
    (79, 48, 4, 333),         # a depth and a width no fused kernel has; nothing is a multiple of the tile sizes
    (5, 7, 2, 1),             # one sample, one hidden layer
    (33, 200, 3, 130),        # wider than three column tiles
    (73, 256, 5, 60000),      # three chunks of the sample axis (25 600 rows each): slots accumulate across chunks
    (73, 64, 3, 140000),      # two chunks at the largest chunk size
]


@pytest.mark.parametrize("cin,hidden,nl,n", CASES)
def test_general_decoder_forward_backward_matches_the_oracle(dev, cin, hidden, nl, n):
    from neural_image_compression_v2_amd import fused
    from neural_image_compression_v2_amd.image_compression import ColorDecoder
    g = torch.Generator().manual_seed(100 + cin + hidden + nl)
    mlp = O.init_mlp(cin, hidden, generator=g, n_linear=nl)
    x = (torch.rand(n, cin, generator=g) - 0.5) * 2.0
    dy = torch.randn(n, 3, generator=g)
    y_ref, dx_ref, g_ref = oracle_fwd_bwd(x, mlp, dy)
    assert not fused.decoder_is_specialised(cin, hidden, nl) or (cin, hidden, nl) == (73, 64, 3)
    dec = ColorDecoder(cin, hidden, nl).to(dev)
    with torch.no_grad():
        for p, t in zip(dec.linear_params(), mlp.tensors()):
            p.copy_(t)
    xg = x.to(dev).requires_grad_(True)
    if (cin, hidden, nl) == (73, 64, 3):
        # force the general kernels (the module would pick the MFMA kernel for this shape)
        y, dx, grads = _general_direct(dev, xg.detach(), [p.detach() for p in dec.linear_params()], dy.to(dev))
    else:
        y = dec(xg)
        (y * dy.to(dev)).sum().backward()
        dx, grads = xg.grad, [p.grad for p in dec.linear_params()]
    torch.cuda.synchronize()
    tol = 2e-5
    assert relmax(y, y_ref) < tol
    assert relmax(dx, dx_ref) < tol
    for i, (a, b) in enumerate(zip(grads, g_ref)):
        assert relmax(a, b) < tol, (i, relmax(a, b))


def _general_direct(dev, x, params, dy):
    """nic_decoder_general_forward / _backward through ctypes, whatever the shape"""
    from neural_image_compression_v2_amd import _lib, fused
    lib = _lib.load()
    n, cin = x.shape
    hidden, nl = params[0].shape[0], len(params) // 2
    m = fused._mlp_struct(params)
    y = torch.empty(n, 3, device=dev)
    ws = torch.empty(int(lib.nic_decoder_general_workspace_bytes(n, cin, hidden, nl, 1)), dtype=torch.uint8, device=dev)
    _lib.check(lib.nic_decoder_general_forward(ctypes.byref(m), _lib.ptr(x), n, cin, hidden, _lib.ptr(y), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
    dx = torch.empty_like(x)
    gm = [torch.empty_like(p) for p in params]
    gs = fused._grads_struct(gm)
    _lib.check(lib.nic_decoder_general_backward(ctypes.byref(m), _lib.ptr(x), _lib.ptr(dy), n, cin, hidden, _lib.ptr(dx), ctypes.byref(gs), _lib.ptr(ws),
                                                ws.numel(), _lib.stream_ptr(dev)))
    return y, dx, gm


def test_general_decoder_against_the_mfma_decoder_and_run_to_run(dev):
    """same inputs through the specialised decoder kernel (fp32 MFMA) and the general kernels: two implementations, one result; the general
    route is bit-stable run to run (fixed summation order)"""
    from neural_image_compression_v2_amd.image_compression import ColorDecoder
    torch.manual_seed(3)
    dec = ColorDecoder(79, 64, 3).to(dev)
    n = 50000
    x = (torch.rand(n, 79, device=dev) - 0.5)
    dy = torch.randn(n, 3, device=dev)
    xg = x.clone().requires_grad_(True)
    y = dec(xg)
    (y * dy).sum().backward()
    params = [p.detach() for p in dec.linear_params()]
    y2, dx2, g2 = _general_direct(dev, x, params, dy)
    y3, dx3, g3 = _general_direct(dev, x, params, dy)
    assert relmax(y2, y) < 1e-5 and relmax(dx2, xg.grad) < 1e-5
    for a, p in zip(g2, dec.linear_params()):
        assert relmax(a, p.grad) < 2e-5
    assert torch.equal(y2, y3) and torch.equal(dx2, dx3) and all(torch.equal(a, b) for a, b in zip(g2, g3))


def test_general_decoder_argument_checks(dev):
    from neural_image_compression_v2_amd import _lib, fused
    lib = _lib.load()
    assert lib.nic_decoder_general_workspace_bytes(100, 73, 64, 6, 1) == 0           # depth beyond NIC_MAX_LINEAR
    assert lib.nic_decoder_general_workspace_bytes(100, 0, 64, 3, 1) == 0
    x = torch.zeros(10, 20, device=dev)
    params = [t.to(dev) for t in O.init_mlp(20, 16, n_linear=3).tensors()]
    m = fused._mlp_struct(params)
    y = torch.empty(10, 3, device=dev)
    small = torch.empty(16, dtype=torch.uint8, device=dev)
    rc = lib.nic_decoder_general_forward(ctypes.byref(m), _lib.ptr(x), 10, 20, 16, _lib.ptr(y), _lib.ptr(small), small.numel(), _lib.stream_ptr(dev))
    assert rc == -4                                                                  # NIC_E_WORKSPACE
    rc = lib.nic_decoder_general_forward(ctypes.byref(m), None, 10, 20, 16, _lib.ptr(y), _lib.ptr(small), small.numel(), _lib.stream_ptr(dev))
    assert rc == -1                                                                  # NIC_E_NULL
    # an empty batch is a no-op forward
    ws = torch.empty(int(lib.nic_decoder_general_workspace_bytes(0, 20, 16, 3, 0)), dtype=torch.uint8, device=dev)
    assert lib.nic_decoder_general_forward(ctypes.byref(m), _lib.ptr(x), 0, 20, 16, _lib.ptr(y), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)) == 0


def _image(S, D):
    u = torch.linspace(0, 1, S)
    if D == 2:
        img = torch.stack([0.5 + 0.25 * torch.sin(6.28 * (c + 1) * u)[:, None] * torch.cos(6.28 * (c + 2) * u)[None, :] for c in range(3)])
    else:
        img = torch.stack([0.5 + 0.25 * torch.sin(6.28 * (c + 1) * u)[:, None, None] * torch.cos(6.28 * u)[None, :, None] * torch.cos(3.14 * u)[None, None, :]
                           for c in range(3)])
    return img.clamp(0, 1)


FLAG_CASES = [
    dict(IMAGE_SIZE=512, HIDDEN_LAYER_CHANNELS=96),                                                      # wider than the fused kernels' 64
    dict(IMAGE_SIZE=512, HIDDEN_LAYER_CHANNELS=128, FEATURE_PYRAMID_CHANNELS=8, PE_CHANNELS=4, TF_USE_TRI_PE=False),
    dict(IMAGE_SIZE=512, DECODER_LINEAR_LAYERS=4, FEATURE_PYRAMID_CHANNELS=20),                          # a depth and a channel count outside every list
    dict(IMAGE_SIZE=32, IMAGE_DIMENSION=3, COMPRESSION_METHOD=4, CROP_MIP_LEVEL=4, HIDDEN_LAYER_CHANNELS=96),
    dict(IMAGE_SIZE=32, IMAGE_DIMENSION=3, COMPRESSION_METHOD=3, CROP_MIP_LEVEL=4, HIDDEN_LAYER_CHANNELS=80, FEATURE_PYRAMID_CHANNELS=4),
]


@pytest.mark.parametrize("flags", FLAG_CASES, ids=lambda f: ",".join(f"{k}={v}" for k, v in f.items() if k != "IMAGE_SIZE"))
def test_host_loop_on_flags_without_a_fused_kernel(dev, flags):
    """var2.py:68-72 are command-line flags of the reference: any value must run.  (a) one noise-free step of the layer-wise route against the
    oracle's autograd of the same composition (encode -> decoder -> MSE): loss, grid gradients, decoder gradients; (b) the product loop
    falls back by itself (nothing refused), trains (loss falls) and its decode equals the oracle's decode of the final state."""
    import random
    from neural_image_compression_v2_amd.image_compression import ImageCompression
    from neural_image_compression_v2_amd.var2 import Settings
    cfg = Settings(NUM_EPOCHS=30, NUM_CROPS=2, TF_NO_MIP=True, **flags)
    D, S = cfg.FP_DIMENSION, cfg.IMAGE_SIZE
    img = _image(S, D)
    den = 255.0 if D == 2 else 256.0
    codes = torch.round(img * (den - 1)).to(torch.uint8)
    ic = ImageCompression(cfg, dev, seed=0)
    ic.set_images([codes], den=den)
    C, P, H, NL = cfg.FEATURE_PYRAMID_CHANNELS, cfg.PE_CHANNELS, cfg.HIDDEN_LAYER_CHANNELS, cfg.DECODER_LINEAR_LAYERS
    method = cfg.COMPRESSION_METHOD if D == 3 else 1
    crop = ic.train_sample_number(0)
    # ---- (a) one step, no noise
    fp = ic.feature_pyramid
    coord = [[3, 5, 2][:D], [S - crop, 0, S - crop][:D]]
    if D == 2:
        x = ic.create_decoder_input_2d(fp, coord, 2, 0, 0)
    elif method == 4:
        x = ic.create_decoder_input_3d_v2(fp, coord, 2, 0, 0)
    else:
        x = ic.create_decoder_input_3d(fp, coord, 2, 0, 0)
    assert x.shape == (2 * crop ** D, cfg.DECODER_INPUT_CHANNELS)
    target = torch.rand(x.shape[0], 3, generator=torch.Generator().manual_seed(4))
    y = ic.decoder(x)
    loss = ((y - target.to(dev)) ** 2).mean()
    loss.backward()
    g0r, g1r = fp[0].detach().cpu().clone().requires_grad_(True), fp[1].detach().cpu().clone().requires_grad_(True)
    mlp_ref = O.MLPParams.from_state_dict({k: v.detach().cpu() for k, v in ic.decoder.state_dict().items()}).requires_grad_(True)
    assert len(mlp_ref.w) == NL and mlp_ref.w[0].shape == (H, cfg.DECODER_INPUT_CHANNELS)
    xr = O.create_decoder_input(g0r, g1r, coord, (crop,) * D, 0.25, 0, P, method=method, use_tri_pe=cfg.TF_USE_TRI_PE)
    assert relmax(x, xr) < 1e-6
    loss_r = torch.nn.functional.mse_loss(O.mlp_forward(xr, mlp_ref), target)
    loss_r.backward()
    assert abs(float(loss.detach()) - float(loss_r.detach())) < 1e-5 * float(loss_r.detach())
    assert relmax(fp[0].grad, g0r.grad) < 1e-4 and relmax(fp[1].grad, g1r.grad) < 1e-4
    for p, r in zip(ic.decoder.linear_params(), mlp_ref.tensors()):
        assert relmax(p.grad, r.grad) < 1e-4
    ic.optimizer.zero_grad()
    # ---- (b) the loop
    torch.manual_seed(1)
    random.seed(1)
    p0 = float(ic.psnr(ic.feature_pyramid))
    fp2 = ic.train_models(ic.feature_pyramid)
    assert ic._no_fused_kernel and ic._no_fused_decode
    losses = torch.stack(ic.loss_history).cpu()
    assert bool(torch.isfinite(losses).all()) and float(losses[-10:].mean()) < float(losses[:5].mean()), losses
    assert float(ic.psnr(fp2)) > p0
    rec = ic.decode_image(fp2, ic.decoder, 0)
    mlp_fin = O.MLPParams.from_state_dict({k: v.detach().cpu() for k, v in ic.decoder.state_dict().items()})
    xr = O.create_decoder_input(fp2[0].detach().cpu(), fp2[1].detach().cpu(), [[0] * D], (S,) * D, 0.25, 0, P, method=method,
                                use_tri_pe=cfg.TF_USE_TRI_PE)
    rec_r = O.mlp_forward(xr, mlp_fin).reshape(*([S] * D), 3)
    assert relmax(rec, rec_r) < 2e-5


# ------------------------------------------------------------------------------------------------------
# multi-level mode (neural_image_compression_v2_amd/multilevel.py): no reference semantics - pinned through the shared primitives
# ------------------------------------------------------------------------------------------------------
def _oracle_multilevel_input(fp, coord, extent, levels, pe_channels, use_tri_pe):
    cols = []
    for l in range(levels):
        e = O.create_decoder_input(fp[2 * l], fp[2 * l + 1], coord, extent, 2.0 ** (-2 * (l + 1)), 0, pe_channels, use_tri_pe=use_tri_pe)
        cols.append(e if l == levels - 1 else e[:, :-1])
    return torch.cat(cols, dim=1)


@pytest.mark.parametrize("size,levels,C,P,H,NL,tri", [((256, 256), 3, 4, 6, 32, 3, True), ((200, 136), 2, 12, 6, 64, 5, False), ((64, 64), 1, 12, 6, 64, 3, True)])
def test_multilevel_field_matches_the_oracle_composition(dev, size, levels, C, P, H, NL, tri):
    """x = [enc_0 | .. | enc_{L-1} | lod] with enc_l the reference's encoding of pair l at step 4^-(l+1): decoder input, output, loss and the
    gradients of all 2 L grids and the decoder against autograd through the oracle's create_decoder_input / mlp_forward composition; the
    grids follow ceil(S / cell) + 1 (the reference's sizes on power-of-two squares); L = 1 is the reference's own single-pair input."""
    from neural_image_compression_v2_amd.multilevel import MultiLevelField, level_nodes, max_levels
    f = MultiLevelField(size, levels, channels=C, pe_channels=P, hidden=H, n_linear=NL, device=dev, use_tri_pe=tri, seed=3)
    assert len(f.fp) == 2 * levels and f.cin == levels * (5 * C + 2 * P) + 1
    if size == (256, 256):
        assert [tuple(g.shape[1:]) for g in f.fp] == [(65, 65), (33, 33), (17, 17), (9, 9), (5, 5), (3, 3)]       # = create_pyramid(64, ..) of the reference
        assert max_levels(size) == 3 and max_levels((3840, 2160)) == 5
    ext = (48, 40)
    coord = [[3, 5], [size[0] - ext[0], size[1] - ext[1]]]                  # the second crop touches the far edge of every level
    g = torch.Generator().manual_seed(9)
    target = torch.rand(2 * ext[0] * ext[1], 3, generator=g)
    x = f.decoder_input(coord, ext)
    y = f.decoder(x)
    loss = ((y - target.to(dev)) ** 2).mean()
    loss.backward()
    fp_ref = [t.detach().cpu().clone().requires_grad_(True) for t in f.fp]
    mlp_ref = O.MLPParams.from_state_dict({k: v.detach().cpu() for k, v in f.decoder.state_dict().items()}).requires_grad_(True)
    xr = _oracle_multilevel_input(fp_ref, coord, ext, levels, P, tri)
    assert xr.shape == x.shape and relmax(x, xr) < 1e-6
    if levels == 1:
        assert torch.equal(x.detach().cpu(), O.create_decoder_input(fp_ref[0], fp_ref[1], coord, ext, 0.25, 0, P, use_tri_pe=tri).detach())
    yr = O.mlp_forward(xr, mlp_ref)
    loss_r = torch.nn.functional.mse_loss(yr, target)
    loss_r.backward()
    assert relmax(y, yr) < 1e-5 and abs(float(loss.detach()) - float(loss_r.detach())) < 1e-5 * float(loss_r.detach())
    for i, (a, b) in enumerate(zip(f.fp, fp_ref)):
        assert relmax(a.grad, b.grad) < 1e-4, (i, relmax(a.grad, b.grad))
    for p, r in zip(f.decoder.linear_params(), mlp_ref.tensors()):
        assert relmax(p.grad, r.grad) < 1e-4


def test_multilevel_field_trains_and_decodes(dev):
    """a 60-step fit of a 256 x 192 image with 3 pairs: chunked whole-image passes (gradients accumulated over the chunks, one optimiser step) -
    the loss falls, the grids stay inside the quantiser's range, decode() equals the oracle composition on the final state"""
    from neural_image_compression_v2_amd.multilevel import MultiLevelField
    S = (256, 192)
    u, v = torch.linspace(0, 1, S[0]), torch.linspace(0, 1, S[1])
    img = torch.stack([0.5 + 0.25 * torch.sin(6.28 * (c + 1) * u)[:, None] * torch.cos(6.28 * (c + 2) * v)[None, :] for c in range(3)]).clamp(0, 1)
    f = MultiLevelField(S, 3, hidden=64, n_linear=3, device=dev, seed=0)
    f.set_schedule(60)
    tgt = img.permute(1, 2, 0).to(dev)                                       # [S_x, S_y, 3]
    losses = []
    for step in range(60):
        total = 0.0
        chunks = [(0, 0), (128, 0)]
        for k, (x0, y0) in enumerate(chunks):
            t = tgt[x0:x0 + 128, y0:y0 + 192].reshape(-1, 3)
            total += float(f.train_step([[x0, y0]], (128, 192), t, accumulate=k > 0, scale=1.0 / len(chunks), step=k == len(chunks) - 1))
        losses.append(total)
    assert losses[-1] < 0.3 * losses[0], losses
    lo = -(2 ** 8 - 1) / 2 ** 9
    assert all(float(g.detach().min()) >= lo and float(g.detach().max()) <= 0.5 for g in f.fp)
    rec = f.decode(tile=100)                                                 # ragged tiles
    mlp_fin = O.MLPParams.from_state_dict({k: v.detach().cpu() for k, v in f.decoder.state_dict().items()})
    xr = _oracle_multilevel_input([g.detach().cpu() for g in f.fp], [[0, 0]], S, 3, 6, True)
    assert relmax(rec, O.mlp_forward(xr, mlp_fin).reshape(S[0], S[1], 3)) < 2e-5
    mse = float(((rec - tgt) ** 2).mean())
    assert mse < 0.3 * float(((0.5 - tgt) ** 2).mean())


# ------------------------------------------------------------------------------------------------------
# hipGraph-captured training loop (ImageCompression.train_models_graph; nic_sampler_step_begin, nic_fused_forward_backward_img_dev, nic_adam_multi_dev)
# ------------------------------------------------------------------------------------------------------
GRAPH_CASES = [
    dict(IMAGE_SIZE=512, NUM_CROPS=4),                                                             # 2D, split-bf16 products: fused_train16
    dict(IMAGE_SIZE=512, NUM_CROPS=4, TF_PLAIN_BF16=1),                                            # 2D, plain bf16: fused_q16
    dict(IMAGE_SIZE=64, IMAGE_DIMENSION=3, COMPRESSION_METHOD=4, CROP_MIP_LEVEL=5, NUM_CROPS=8, TF_PLAIN_BF16=1),   # the reference's sweep shape, method 4
    dict(IMAGE_SIZE=64, IMAGE_DIMENSION=3, COMPRESSION_METHOD=3, CROP_MIP_LEVEL=5, NUM_CROPS=8, MLP_NUM_DTYPE=16, TF_PLAIN_BF16=1),   # method 3, float16 grid storage
    dict(IMAGE_SIZE=512, NUM_CROPS=2, TF_PLAIN_BF16=1, DECODER_LINEAR_LAYERS=5, TF_GRID_BF16=True),   # the north star's decoder and storage
]


@pytest.mark.parametrize("flags", GRAPH_CASES, ids=lambda f: ",".join(f"{k}={v}" for k, v in f.items()))
def test_graph_captured_loop_matches_the_host_loop(dev, flags):
    """the same fit twice - train_models (host loop, device sampler) and train_models_graph (captured replays: origins, noise offset, Adam
    step and cosine learning rate all taken from device memory): same origins, same noise, same learning rates, so the loss of every step,
    the final grids, decoder and optimiser state agree to the order of the atomic sums; the tail after 0.95 NUM_EPOCHS runs in both"""
    from neural_image_compression_v2_amd.image_compression import ImageCompression
    from neural_image_compression_v2_amd.var2 import Settings
    res = {}
    for mode in ("host", "graph"):
        cfg = Settings(NUM_EPOCHS=50, TF_NO_MIP=True, TF_DEVICE_SAMPLER=True, SAMPLER_SEED=11, **flags)
        D, S = cfg.FP_DIMENSION, cfg.IMAGE_SIZE
        den = 255.0 if D == 2 else 256.0
        ic = ImageCompression(cfg, dev, seed=0)
        ic.set_images([torch.round(_image(S, D) * (den - 1)).to(torch.uint8)], den=den)
        if mode == "host":
            fp = ic.train_models(ic.feature_pyramid)
        else:
            fp = ic.train_models_graph(ic.feature_pyramid, steps_per_graph=8)
        torch.cuda.synchronize()
        assert ic.step_count == 50 and len(ic.loss_history) == 50
        st = ic.optimizer.state
        res[mode] = dict(loss=torch.stack([l.reshape(()) for l in ic.loss_history]).cpu(), fp=[g.detach().cpu() for g in ic.feature_pyramid],
                         dec=[p.detach().cpu() for p in ic.decoder.linear_params()], lr=[g["lr"] for g in ic.optimizer.param_groups],
                         steps=[int(st[p]["step"].item()) for p in ic.feature_pyramid], m=[st[p]["exp_avg"].cpu() for p in ic.feature_pyramid],
                         psnr=float(ic.psnr(fp)))
    h, g = res["host"], res["graph"]
    assert h["lr"] == g["lr"] and h["steps"] == g["steps"]                  # bit-identical schedule, same step counts (grids stop at the freeze)
    assert bool(torch.isfinite(g["loss"]).all()) and float(g["loss"][-5:].mean()) < float(g["loss"][:5].mean())
    rel = float(((h["loss"] - g["loss"]).abs() / h["loss"]).max())
    assert rel < 2e-3, rel                                                  # atomic summation order only (measured ~1e-5; bf16 modes amplify it)
    for a, b in zip(h["fp"] + h["dec"] + h["m"], g["fp"] + g["dec"] + g["m"]):
        assert relmax(b, a) < 5e-3, relmax(b, a)
    assert abs(h["psnr"] - g["psnr"]) < 0.02


def test_step_begin_and_adam_dev_entry_points(dev):
    """nic_sampler_step_begin: counters, loss filing and the origins of nic_sampler_draw_origins; nic_adam_multi_dev against nic_adam_multi
    step by step (schedule rows = what the host call forms per step: bit-identical parameters)"""
    from neural_image_compression_v2_amd import _lib
    from neural_image_compression_v2_amd.optim import CosineAnnealing, FusedAdam
    from neural_image_compression_v2_amd.sampler import DeviceSampler
    lib = _lib.load()
    counters = torch.tensor([5, -1], dtype=torch.int64, device=dev)
    org = torch.zeros(8 * 3, dtype=torch.int32, device=dev)
    slot = torch.tensor([0.25], device=dev)
    hist = torch.zeros(10, device=dev)
    _lib.check(lib.nic_sampler_step_begin(123, _lib.ptr(counters), 8, 3, 33, _lib.ptr(org), _lib.ptr(slot), _lib.ptr(hist), 10, _lib.stream_ptr(dev)))
    assert counters.tolist() == [6, 5] and hist.tolist() == [0, 0, 0, 0, 0.25, 0, 0, 0, 0, 0]
    smp = DeviceSampler(123, dev)
    assert torch.equal(org.view(8, 3).cpu(), smp.origins_host(5, 8, 3, 33))
    # Adam: two groups, 12 steps, clamp on the first group
    torch.manual_seed(0)
    def make():
        a = torch.nn.Parameter(torch.rand(5000, device=dev) - 0.5)
        b = torch.nn.Parameter(torch.randn(64, 73, device=dev))
        opt = FusedAdam([{"params": [a], "lr": 0.01}, {"params": [b], "lr": 0.005}])
        opt.set_clamp([a], -0.498, 0.5)
        return a, b, opt, CosineAnnealing(opt, T_max=20)
    torch.manual_seed(1); a1, b1, o1, s1 = make()
    torch.manual_seed(1); a2, b2, o2, s2 = make()
    ga, gb = torch.zeros_like(a1), torch.zeros_like(b1)
    a2.grad, b2.grad = ga, gb
    adam = o2.dev_table([(a2, ga), (b2, gb)], 3, s2.peek(12))
    ctr = torch.tensor([3], dtype=torch.int64, device=dev)
    gen = torch.Generator(device=dev).manual_seed(4)
    for k in range(12):
        ga.copy_(torch.randn(5000, device=dev, generator=gen)); gb.copy_(torch.randn(64, 73, device=dev, generator=gen))
        a1.grad, b1.grad = ga.clone(), gb.clone()
        o1.step(); s1.step()
        adam.launch(ctr.data_ptr())
        ctr += 1
    adam.commit(12); s2.advance(12)
    assert torch.equal(a1.detach(), a2.detach()) and torch.equal(b1.detach(), b2.detach())
    assert [g["lr"] for g in o1.param_groups] == [g["lr"] for g in o2.param_groups]
    assert int(o2.state[a2]["step"].item()) == 12


def test_rgbx_resize_is_the_references_mip_chain(dev):
    """nic_rgbx_resample_axis + sampler.resize_coeffs against the golden levels Pillow made of the reference's sample image (tests/golden/mipchain.npz),
    against the oracle's restatement on non-square / non-integer ratios, and through Settings.TF_MIP_FILTER in the host loop's RGBX pyramid"""
    import os
    from neural_image_compression_v2_amd.sampler import build_rgbx_pyramid, rgbx_interleave, rgbx_resize
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mipchain.npz"))

    def unpack(t):
        w = t.cpu().numpy().astype(np.uint32)
        return np.stack([(w >> (8 * c)) & 255 for c in range(3)], axis=-1).astype(np.uint8)

    img = torch.from_numpy(g["image"]).permute(2, 0, 1).contiguous().to(dev)              # [3, 128, 128] codes
    pyr = build_rgbx_pyramid(img, 8)                                                        # default filter: the reference's
    for i in range(1, 8):
        assert np.array_equal(unpack(pyr[i].image), g[f"level_{i}"]), i
    rng = np.random.default_rng(3)
    for (H, W, oh, ow) in [(96, 160, 48, 80), (100, 60, 33, 17), (37, 53, 37, 20), (64, 64, 1, 1), (512, 512, 2, 2)]:
        a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        lvl = rgbx_interleave(torch.from_numpy(a).permute(2, 0, 1).contiguous().to(dev))
        assert np.array_equal(unpack(rgbx_resize(lvl, oh, ow)), O.pil_resize_bilinear(a, oh, ow)), (H, W, oh, ow)



# ------------------------------------------------------------------------------------------------------
# HIDDEN_LAYER_CHANNELS below 64 on the FUSED kernels: zero-padded parameters (fused.PaddedMlp, nic_decoder_pad / _unpad)
# ------------------------------------------------------------------------------------------------------
PAD_CASES = [
    # dim, method, hidden, n_linear, mode kwargs, tolerance (y, grads)
    (2, 1, 32, 3, dict(), 1e-5, 2e-5),                              # fp32 MFMA kernel
    (2, 1, 32, 3, dict(split_bf16=True), 2e-5, 5e-5),               # fused_train16
    (2, 1, 48, 5, dict(split_bf16=True), 2e-5, 1e-4),               # fused_mlpn, 5 layers
    (2, 1, 32, 3, dict(bf16=True), None, None),                     # fused_q16 (held to the emulating oracle)
    (3, 3, 16, 3, dict(), 1e-5, 2e-5),
    (3, 4, 32, 3, dict(bf16=True), None, None),
    (3, 4, 1, 3, dict(), 1e-5, 2e-5),                               # a single hidden unit
]


@pytest.mark.parametrize("dim,method,H,NL,kw,tol_y,tol_g", PAD_CASES, ids=lambda v: str(v) if not isinstance(v, dict) else ",".join(v) or "fp32")
def test_fused_kernels_serve_narrower_decoders_by_zero_padding(dev, dim, method, H, NL, kw, tol_y, tol_g):
    """fused step and fused decode with HIDDEN_LAYER_CHANNELS = H < 64 against the oracle's H-wide decoder: outputs, loss, grid gradients and all
    decoder gradients (shapes [H, ..]: the padded rows never reach the caller); plain-bf16 cases against the precision-emulating oracle"""
    from neural_image_compression_v2_amd import _lib, fused
    g = torch.Generator().manual_seed(50 + H + NL + dim)
    base = 16
    fp, _ = O.create_pyramid(base, 12, 8, no_mip=True, generator=g, dim=dim)
    g0, g1 = fp[0].detach(), fp[1].detach()
    cin = O.decoder_input_channels(12, 6, dim, method)
    mlp = O.init_mlp(cin, H, generator=g, n_linear=NL)
    extent = (24, 20) if dim == 2 else (12, 8, 10)
    origins = [[3, 5, 2][:dim], [30, 0, 17][:dim]]
    n = 2 * int(np.prod(extent))
    target = torch.rand(n, 3, generator=g)
    noise = O.kernel_noise(n, cin, 8, seed=11, offset=5, quarter=bool(kw.get("bf16")))
    emulate = "bf16" if kw.get("bf16") else None
    ref = O.forward_backward(g0, g1, mlp, origins, extent, 0.25, 0, target, noise, 6, method=method, **({"emulate": emulate} if emulate else {}))
    geo = fused.PathGeometry(dim=dim, method=method, step_number=0.25, mip_level=0, extent=extent, num_crops=2, hidden=H,
                             noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=11, noise_offset=5, **kw)
    params = [t.to(dev) for t in mlp.tensors()]
    out = fused.fused_forward_backward(geo, g0.to(dev), g1.to(dev), origins, params, target.to(dev), want_y=True)
    ty, tg = (tol_y, tol_g) if tol_y is not None else (2e-3, 3e-3)
    assert relmax(out.y, ref.y) < ty and relmax(out.loss.reshape(()), torch.as_tensor(ref.loss).reshape(())) < max(ty, 1e-5)
    assert relmax(out.grad_g0, ref.grad_g0) < tg and relmax(out.grad_g1, ref.grad_g1) < tg
    for i, (a, b) in enumerate(zip(out.grad_mlp, ref.grad_mlp)):
        assert tuple(a.shape) == tuple(b.shape) and relmax(a, b) < tg, (i, relmax(a, b))
    # decode (no noise): the forward kernels
    geo_d = fused.PathGeometry(dim=dim, method=method, step_number=0.25, mip_level=0, extent=extent, num_crops=2, hidden=H, **kw)
    y = fused.fused_forward(geo_d, g0.to(dev), g1.to(dev), origins, params)
    x = O.create_decoder_input(g0, g1, origins, extent, 0.25, 0, 6, method=method)
    assert relmax(y, O.mlp_forward(x, mlp)) < (5e-3 if kw.get("bf16") else 2e-5)


def test_host_loop_with_32_hidden_units_stays_on_the_fused_kernels(dev):
    """HIDDEN_LAYER_CHANNELS = 32 (var2.py:72): the product loop keeps the fused step and decode (no fallback), trains, and a captured run equals it"""
    from neural_image_compression_v2_amd.image_compression import ImageCompression
    from neural_image_compression_v2_amd.var2 import Settings
    res = {}
    for mode in ("host", "graph"):
        cfg = Settings(IMAGE_SIZE=512, NUM_EPOCHS=40, NUM_CROPS=2, TF_NO_MIP=True, HIDDEN_LAYER_CHANNELS=32, TF_DEVICE_SAMPLER=True)
        ic = ImageCompression(cfg, dev, seed=0)
        ic.set_images([torch.round(_image(512, 2) * 255).to(torch.uint8)])
        p0 = float(ic.psnr(ic.feature_pyramid))
        fp = ic.train_models(ic.feature_pyramid) if mode == "host" else ic.train_models_graph(ic.feature_pyramid, steps_per_graph=4)
        assert not getattr(ic, "_no_fused_kernel", False) and not getattr(ic, "_no_fused_decode", False)
        assert ic.decoder.decoder[0].weight.shape == (32, 73) and ic.decoder.decoder[0].weight.grad.shape == (32, 73)
        res[mode] = (torch.stack([l.reshape(()) for l in ic.loss_history]).cpu(), float(ic.psnr(fp)))
        assert res[mode][1] > p0 + 1.0
    rel = float(((res["host"][0] - res["graph"][0]).abs() / res["host"][0]).max())
    assert rel < 2e-3 and abs(res["host"][1] - res["graph"][1]) < 0.02
