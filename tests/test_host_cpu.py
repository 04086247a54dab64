"""CPU-only tests (run with -m "not gpu"): the C-ABI library loads and exports every symbol the header declares, the
ctypes mirror of nic_path_desc matches the C layout (checked against gcc), host logic (settings, level maps, geometry
validation, gradient bucket layout), no product path runs without a HIP device, and the data-parallel sharding /
bucket all-reduce logic over gloo with the CPU oracle injected as the step function."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nicv2_hip.h")


@pytest.fixture(scope="module")
def lib():
    from neural_image_compression_v2_amd import _build, _lib
    if not os.path.exists(_lib.LIB_PATH):
        _build.build(verbose=False)
    return _lib.load()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nic_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    from neural_image_compression_v2_amd import _lib
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nicv2_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES and the header disagree"
    assert lib.nic_abi_version() == _lib.NIC_ABI_VERSION == 9
    assert lib.nic_error_string(-2).decode().startswith("unsupported")
    assert lib.nic_decoder_input_channels(2, 1, 12, 6) == 73          # var2.py:114-118
    assert lib.nic_decoder_input_channels(3, 3, 12, 6) == 127
    assert lib.nic_decoder_input_channels(3, 4, 12, 6) == 79
    assert lib.nic_workspace_bytes(None) > 0                           # no GPU call involved


def test_path_desc_layout_matches_the_c_header():
    from neural_image_compression_v2_amd._lib import NicAdamTensor, NicMlp, NicMlPairs, NicPathDesc, NicStepTail, NicTargetImage
    fields = [f[0] for f in NicPathDesc._fields_]
    afields = [f[0] for f in NicAdamTensor._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(){",
            'printf("%zu\\n", sizeof(nic_path_desc));']
    prog += [f'printf("%zu\\n", offsetof(nic_path_desc, {f}));' for f in fields]
    prog += ['printf("%zu\\n", sizeof(nic_mlp));', 'printf("%zu\\n", sizeof(nic_adam_tensor));']
    prog += [f'printf("%zu\\n", offsetof(nic_adam_tensor, {f}));' for f in afields]
    tfields = [f[0] for f in NicTargetImage._fields_]
    prog += ['printf("%zu\\n", sizeof(nic_target_image));']
    prog += [f'printf("%zu\\n", offsetof(nic_target_image, {f}));' for f in tfields]
    mfields = [f[0] for f in NicMlPairs._fields_]
    prog += ['printf("%zu\\n", sizeof(nic_ml_pairs));']
    prog += [f'printf("%zu\\n", offsetof(nic_ml_pairs, {f}));' for f in mfields]
    sfields = [f[0] for f in NicStepTail._fields_]
    prog += ['printf("%zu\\n", sizeof(nic_step_tail));']
    prog += [f'printf("%zu\\n", offsetof(nic_step_tail, {f}));' for f in sfields]
    prog += ["return 0;}"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write("\n".join(prog))
        subprocess.run(["gcc", "-std=c11", src, "-o", exe], check=True)
        vals = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    ns = len(sfields)
    assert vals[-1 - ns] == ctypes.sizeof(NicStepTail)
    for f, off in zip(sfields, vals[-ns:]):
        assert getattr(NicStepTail, f).offset == off, f
    vals = vals[:-1 - ns]
    nf = len(fields)
    assert vals[0] == ctypes.sizeof(NicPathDesc)
    for f, off in zip(fields, vals[1:1 + nf]):
        assert getattr(NicPathDesc, f).offset == off, f
    assert vals[1 + nf] == ctypes.sizeof(NicMlp)
    assert vals[2 + nf] == ctypes.sizeof(NicAdamTensor)
    na = len(afields)
    for f, off in zip(afields, vals[3 + nf:3 + nf + na]):
        assert getattr(NicAdamTensor, f).offset == off, f
    assert vals[3 + nf + na] == ctypes.sizeof(NicTargetImage)
    nt = len(tfields)
    for f, off in zip(tfields, vals[4 + nf + na:4 + nf + na + nt]):
        assert getattr(NicTargetImage, f).offset == off, f
    assert vals[4 + nf + na + nt] == ctypes.sizeof(NicMlPairs)
    for f, off in zip(mfields, vals[5 + nf + na + nt:]):
        assert getattr(NicMlPairs, f).offset == off, f


def test_argument_errors_are_reported_before_any_gpu_work(lib):
    """error convention of the C ABI (include/nicv2_hip.h): negative NIC_E_* codes for bad arguments, decided on the host - no
    device is needed (and none is touched) to get them"""
    from neural_image_compression_v2_amd import _lib
    E_NULL, E_UNSUPPORTED, E_SHAPE, E_WORKSPACE, E_ARG = -1, -2, -3, -4, -5
    d = _lib.NicPathDesc()
    d.dim, d.method, d.channels, d.pe_channels, d.hidden = 2, 1, 12, 6, 64
    d.num_crops = 1
    for a in range(3):
        d.extent[a], d.g0_nodes[a], d.g1_nodes[a] = 8, 17, 9
    null = ctypes.c_void_p(0)
    fake = ctypes.c_void_p(16)                              # never dereferenced: the checks fail first
    m = _lib.NicMlp()
    assert lib.nic_fused_forward(None, fake, fake, fake, ctypes.byref(m), null, fake, null) == E_NULL
    assert lib.nic_fused_forward(ctypes.byref(d), null, fake, fake, ctypes.byref(m), null, fake, null) == E_NULL
    assert lib.nic_fused_forward(ctypes.byref(d), fake, fake, fake, ctypes.byref(m), null, fake, null) == E_NULL     # empty nic_mlp
    d.channels = 5
    assert lib.nic_fused_forward(ctypes.byref(d), fake, fake, fake, ctypes.byref(m), null, fake, null) == E_UNSUPPORTED
    d.channels, d.dim, d.method = 12, 3, 1
    assert lib.nic_fused_forward(ctypes.byref(d), fake, fake, fake, ctypes.byref(m), null, fake, null) == E_UNSUPPORTED
    d.dim, d.method, d.num_crops = 2, 1, 0
    for i in range(3):
        m.w[i] = m.b[i] = 16
    assert lib.nic_fused_forward(ctypes.byref(d), fake, fake, fake, ctypes.byref(m), null, fake, null) == E_SHAPE
    d.num_crops, d.log2_step = 1, 40
    assert lib.nic_fused_forward(ctypes.byref(d), fake, fake, fake, ctypes.byref(m), null, fake, null) == E_ARG
    d.log2_step = -2
    gs = _lib.NicMlpGrads()
    assert lib.nic_fused_forward_backward(ctypes.byref(d), fake, fake, fake, ctypes.byref(m), null, fake, null, fake, fake, fake,
                                          ctypes.byref(gs), fake, 16, null) == E_WORKSPACE
    d.passes = 2                                            # repeated passes exist for the training entry points only
    assert lib.nic_fused_forward(ctypes.byref(d), fake, fake, fake, ctypes.byref(m), null, fake, null) == E_ARG
    assert lib.nic_encode(ctypes.byref(d), fake, fake, fake, fake, null) == E_ARG
    assert lib.nic_fused_forward_backward(ctypes.byref(d), fake, fake, fake, ctypes.byref(m), null, fake, null, fake, fake, fake,
                                          ctypes.byref(gs), fake, 16, null) == E_WORKSPACE          # accepted there
    d.passes = -1
    assert lib.nic_fused_forward_backward(ctypes.byref(d), fake, fake, fake, ctypes.byref(m), null, fake, null, fake, fake, fake,
                                          ctypes.byref(gs), fake, 16, null) == E_ARG
    d.passes = 0
    d.noise_mode = _lib.NIC_NOISE_KERNEL
    assert lib.nic_fused_forward_u8(ctypes.byref(d), fake, fake, fake, ctypes.byref(m), fake, null, null) == E_ARG    # decoding adds no noise
    assert lib.nic_adam_multi(None, 3, 0.9, 0.999, 1e-8, null) == E_NULL
    arr = (_lib.NicAdamTensor * 1)()
    assert lib.nic_adam_multi(arr, 33, 0.9, 0.999, 1e-8, null) == E_ARG
    assert lib.nic_adam_multi(arr, 0, 0.9, 0.999, 1e-8, null) == 0
    assert lib.nic_quantize(null, fake, 4, 8, null) == E_NULL
    assert lib.nic_load4fp_u8(fake, fake, 4, 9, null) == E_ARG
    assert b"unsupported" in lib.nic_error_string(E_UNSUPPORTED).lower() or len(lib.nic_error_string(E_UNSUPPORTED)) > 0


def test_no_cpu_path():
    """every product entry point refuses CPU tensors instead of computing something else"""
    from neural_image_compression_v2_amd import fp_def, fused, models, utils
    from neural_image_compression_v2_amd.image_compression import ColorDecoder, ImageCompression
    x = torch.zeros(4, 73)
    for fn in (lambda: models.quantize4fp(x, 8), lambda: models.save4fp(x, 8), lambda: models.quantize_to_bit(x),
               lambda: utils.triangular_positional_encoding(torch.zeros(2, 5), 6), lambda: utils.calculate_psnr(x, x),
               lambda: ColorDecoder()(x), lambda: fp_def.fp_quantize_clamp([x, x], 0, 8)):
        with pytest.raises(RuntimeError):
            fn()
    geo = fused.PathGeometry(2, 1, 0.25, 0, (8, 8), 1)
    with pytest.raises(RuntimeError):
        fused.encode(geo, torch.zeros(12, 17, 17), torch.zeros(12, 9, 9), [(0, 0)])
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            ImageCompression()


def test_oracle_is_not_imported_by_the_product():
    pkg = os.path.join(ROOT, "neural_image_compression_v2_amd")
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b|from\s+\.+oracle\b)|nic_oracle\s*(as|\.|import)", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f"{f} imports the oracle"


def test_settings_and_level_maps():
    from neural_image_compression_v2_amd import fp_def
    from neural_image_compression_v2_amd.var2 import Settings
    s = Settings.from_argv(["IMAGE_PATH=data/x.npy", "FP_BITS=4", "NUM_EPOCHS=320000", "COMPRESSION_METHOD=4", "IMAGE_DIMENSION=3",
                            "IMAGE_SIZE=64", "CROP_MIP_LEVEL=5", "TF_USE_TRI_PE=False", "UNIFORM_DISTRIBUTION_RATE=0.1"])   # the .bat sweeps
    assert (s.FP_BITS, s.NUM_EPOCHS, s.COMPRESSION_METHOD, s.IMAGE_SIZE, s.CROP_MIP_LEVEL) == (4, 320000, 4, 64, 5)
    assert s.TF_USE_TRI_PE is False and s.UNIFORM_DISTRIBUTION_RATE == 0.1 and s.IMAGE_PATH == "data/x.npy"
    assert s.DECODER_INPUT_CHANNELS == 79 and s.FEATURE_PYRAMID_SIZE == 16 and s.CROP_SIZE == 32 and s.MAX_MIP_LEVEL == 0
    assert Settings().DECODER_INPUT_CHANNELS == 73 and Settings(IMAGE_DIMENSION=3, COMPRESSION_METHOD=3).DECODER_INPUT_CHANNELS == 127
    assert Settings(TF_NO_MIP=False).MAX_MIP_LEVEL == 9
    with pytest.raises(ValueError):
        Settings.from_argv(["TF_NO_MIP=maybe"])
    assert dict(fp_def.create_pyramid_mip_levels(512, 128)) == {0: 0, 1: 0, 2: 0, 3: 0, 4: 1, 5: 1, 6: 2, 7: 2, 8: 3, 9: 3}
    assert fp_def.return_pyramid_levels(128) == 4 and fp_def.return_2_power(1024) == 10          # test03.py prints


def test_geometry_and_validation():
    from neural_image_compression_v2_amd import _lib, fused
    g0, g1 = torch.zeros(12, 961, 541), torch.zeros(12, 481, 271)                               # SURVEY 8d config 2 grids
    geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(2160, 3840), num_crops=1)
    d = geo.to_desc(g0, g1)
    assert (d.g0_nodes[0], d.g0_nodes[1], d.g1_nodes[0], d.g1_nodes[1]) == (541, 961, 271, 481)
    assert d.log2_step == -2 and d.g1_weight_mode == _lib.NIC_G1_REFERENCE and abs(d.loss_scale - 1 / (3 * 2160 * 3840)) < 1e-15
    assert geo.cin == 73 and geo.n_samples == 2160 * 3840
    fused.check_origins(geo, torch.tensor([[0, 0]]), g0, g1)
    with pytest.raises(IndexError):
        fused.check_origins(geo, torch.tensor([[1, 0]]), g0, g1)                                # one sample past the grid
    with pytest.raises(IndexError):
        fused.check_origins(geo, torch.tensor([[-1, 0]]), g0, g1)
    # Q6: only step_number == 2 switches the G1 weights off
    for e in range(-3, 4):
        geo2 = fused.PathGeometry(2, 1, pow(2, e), 0, (4, 4), 1)
        assert (geo2.g1_mode() == _lib.NIC_G1_UNWEIGHTED) == (e == 1)
    with pytest.raises(ValueError):
        fused.log2_step_of(3)
    # method fixes the PE family in 3D (fp_def.py:169, 208)
    assert fused.PathGeometry(3, 3, 0.25, 0, (4, 4, 4), 1, use_tri_pe=False).use_tri_pe is True
    assert fused.PathGeometry(3, 4, 0.25, 0, (4, 4, 4), 1, use_tri_pe=True).use_tri_pe is False
    # the sinusoidal divisors are torch's own fp32 values (utils.py:202)
    import math
    ref = torch.exp(torch.arange(0, 6, 2, dtype=torch.float32) * -(math.log(10000.0) / 6))
    assert fused.sinusoidal_div_term(6) == [float(v) for v in ref]
    offs, sizes, total = fused.grad_bucket_layout(geo, g0, g1)
    assert sizes[1:7] == [64 * 73, 64, 64 * 64, 64, 192, 3] and sizes[7] == g0.numel() and all(o % 4 == 0 for o in offs)
    assert total >= sum(sizes)


def test_shard_range_covers_everything():
    from neural_image_compression_v2_amd.distributed import plan_shard, shard_range
    for n in (1, 7, 8, 9, 135):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for s, c in spans:
                assert s == pos
                pos += c
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    p = plan_shard(8, 65536, rank=3, world=8)
    assert (p.crop_start, p.crop_count, p.sample_base, p.n_global) == (3, 1, 3 * 65536, 8 * 65536)


# ------------------------------------------------------------------------------------------------ world_size 2 over gloo
def _oracle_step(geo, g0, g1, org, params, target, **kw):
    """CPU oracle with the product's step signature (test-only backend for distributed.data_parallel_step)"""
    from neural_image_compression_v2_amd import fused
    from oracle import nic_oracle as O
    mlp = O.MLPParams([params[0], params[2], params[4]], [params[1], params[3], params[5]])
    noise = None if geo.noise_mode == 0 else O.kernel_noise(geo.n_samples, geo.cin, geo.num_bits, geo.noise_seed, geo.noise_offset, geo.sample_base)
    r = O.forward_backward(g0, g1, mlp, [tuple(int(v) for v in o) for o in org], geo.extent, geo.step_number, geo.mip_level, target, noise,
                           geo.pe_channels, method=geo.method, use_tri_pe=geo.use_tri_pe, mean_over=(geo.n_samples if geo.loss_scale is None else int(round(1 / (3 * geo.loss_scale)))))
    offs, sizes, total = fused.grad_bucket_layout(geo, g0, g1)
    flat = torch.zeros(total)
    parts = [r.loss.reshape(1)] + [g.reshape(-1) for g in r.grad_mlp] + [r.grad_g0.reshape(-1), r.grad_g1.reshape(-1)]
    for o, part in zip(offs, parts):
        flat[o:o + part.numel()] = part
    return fused.StepOutput(flat[0], None, flat[offs[7]:offs[7] + sizes[7]].view(g0.shape), flat[offs[8]:offs[8] + sizes[8]].view(g1.shape),
                            [flat[offs[1 + i]:offs[1 + i] + sizes[1 + i]] for i in range(6)], flat)


def _dp_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from neural_image_compression_v2_amd import _lib, fused
        from neural_image_compression_v2_amd.distributed import data_parallel_step
        from oracle import nic_oracle as O
        torch.set_num_threads(2)
        g = torch.Generator().manual_seed(1)
        fp, _ = O.create_pyramid(16, 12, 8, no_mip=True, generator=g)
        mlp = O.init_mlp(73, 64, generator=g)
        origins = torch.tensor([[0, 0], [10, 20], [33, 7], [40, 40], [5, 48]])            # 5 crops over 2 ranks: 3 + 2
        geo = fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=(16, 16), num_crops=5,
                                 noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=5, noise_offset=9)
        target = torch.rand(geo.n_samples, 3, generator=g)
        out = data_parallel_step(_oracle_step, geo, fp[0].detach(), fp[1].detach(), origins, mlp.tensors(), target)
        if rank == 0:
            single = _oracle_step(geo, fp[0].detach(), fp[1].detach(), origins, mlp.tensors(), target)
            torch.save({"dp": out.flat, "single": single.flat}, out_path)
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_matches_single_process_gloo():
    """2 ranks, 5 crops: per-rank oracle steps with the global loss scale and global sample ids, ONE all-reduce of the flat
    bucket == the single-process step on all crops (loss, decoder grads, grid grads)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r.pt")
        mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
        r = torch.load(out)
    err = float((r["dp"] - r["single"]).abs().max() / r["single"].abs().max())
    assert err < 1e-6, err


def test_stripe_plans_tile_the_axis():
    from neural_image_compression_v2_amd.distributed import plan_stripes
    for length, cell in ((3840, 8), (1080, 8), (64, 2), (100, 4)):
        for world in (1, 2, 3, 4, 8):
            if length // cell < world:
                continue
            plans = [plan_stripes(length, cell, r, world) for r in range(world)]
            assert plans[0].start == 0 and plans[-1].start + plans[-1].size == length
            for a, b in zip(plans, plans[1:]):
                assert a.start + a.size == b.start and b.start % cell == 0
                # neighbours share exactly one node row per grid: the boundary row
                for level in (0, 1):
                    assert a.node_rows(level)[1] == b.node_rows(level)[0] == a.boundary_rows(level)[a.rank]
    p = plan_stripes(3840, 8, 7, 8)
    assert (p.start, p.size, p.node_rows(0), p.node_rows(1)) == (3360, 480, (840, 960), (420, 480))
    with pytest.raises(ValueError):
        plan_stripes(16, 8, 0, 4)


def _stripe_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from neural_image_compression_v2_amd import _lib, fused
        from neural_image_compression_v2_amd.distributed import assemble_stripes, plan_stripes, stripe_exchange
        from oracle import nic_oracle as O
        torch.set_num_threads(2)
        HH, WW = 16, 32
        g = torch.Generator().manual_seed(3)
        fp, _ = O.create_pyramid((HH // 4, WW // 4), 12, 8, dim=2, no_mip=True, generator=g)          # [12, 9, 5], [12, 5, 3]
        g0, g1 = fp[0].detach().clone(), fp[1].detach().clone()
        mlp = O.init_mlp(73, 64, generator=g)
        image = torch.rand(HH, WW, 3, generator=g)
        plan = plan_stripes(WW, 8, rank, world)
        n_crop = HH * plan.size
        n_global = world * world * n_crop

        def geo_of(num_crops, extent, base):
            return fused.PathGeometry(dim=2, method=1, step_number=0.25, mip_level=0, extent=extent, num_crops=num_crops,
                                      noise_mode=_lib.NIC_NOISE_KERNEL, noise_seed=5, noise_offset=2, sample_base=base,
                                      loss_scale=1.0 / (3.0 * n_global))
        # this rank: `world` crops, all of them its stripe
        geo = geo_of(world, (HH, plan.size), rank * world * n_crop)
        tgt = image[:, plan.start:plan.start + plan.size].reshape(-1, 3).repeat(world, 1)
        out = _oracle_step(geo, g0, g1, torch.tensor([[0, plan.start]] * world), mlp.tensors(), tgt)
        offs, sizes, _ = fused.grad_bucket_layout(geo, g0, g1)
        stripe_exchange(plan, out.flat[:offs[7]], out.grad_g0, out.grad_g1)
        # a plain gradient step on everything this rank holds, then the stripes are put together
        g0 -= 0.5 * out.grad_g0
        g1 -= 0.5 * out.grad_g1
        assemble_stripes(plan, g0, g1)
        # single process: the same crops (stripe r `world` times, ranks in order) with the same global sample ids
        plans = [plan_stripes(WW, 8, r, world) for r in range(world)]
        org_all = torch.tensor([[0, p.start] for p in plans for _ in range(world)])
        tgt_all = torch.cat([image[:, p.start:p.start + p.size].reshape(-1, 3).repeat(world, 1) for p in plans])
        single = _oracle_step(geo_of(world * world, (HH, plan.size), 0), fp[0].detach(), fp[1].detach(), org_all, mlp.tensors(), tgt_all)
        lo0, hi0 = plan.node_rows(0)
        lo1, hi1 = plan.node_rows(1)
        res = {
            "small": float((out.flat[:offs[7]] - single.flat[:offs[7]]).abs().max() / single.flat[:offs[7]].abs().max()),
            "g0": float((out.grad_g0[:, lo0:hi0 + 1] - single.grad_g0[:, lo0:hi0 + 1]).abs().max() / single.grad_g0.abs().max()),
            "g1": float((out.grad_g1[:, lo1:hi1 + 1] - single.grad_g1[:, lo1:hi1 + 1]).abs().max() / single.grad_g1.abs().max()),
            "p0": float((g0 - (fp[0].detach() - 0.5 * single.grad_g0)).abs().max()),
            "p1": float((g1 - (fp[1].detach() - 0.5 * single.grad_g1)).abs().max()),
            "touched_outside": float(out.grad_g0[:, :lo0].abs().sum() + out.grad_g0[:, hi0 + 1:].abs().sum()),
        }
        torch.save(res, out_path + f".{rank}")
    finally:
        dist.destroy_process_group()


def test_stripe_sharded_step_matches_single_process_gloo():
    """2 ranks, grids sharded in stripes of the last sample axis: per-rank oracle steps on the rank's stripe, ONE small all-reduce
    (loss + decoder grads + the boundary node rows) == the single-process step on the same crops, on every row a rank owns; the
    assembled grids after an update == the single-process update."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r.pt")
        mp.spawn(_stripe_worker, args=(2, port, out), nprocs=2, join=True)
        for r in range(2):
            res = torch.load(out + f".{r}")
            assert res["touched_outside"] == 0.0, res
            for k in ("small", "g0", "g1"):
                assert res[k] < 1e-6, (r, res)
            assert res["p0"] < 1e-6 and res["p1"] < 1e-6, (r, res)


def test_integration_md_binding_matches_the_c_struct():
    """the ctypes stub printed in INTEGRATION.md is what a maintainer copies: its nic_path_desc must have the size and field
    offsets of the C struct (a struct 8 bytes short makes check_geometry read garbage)"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class nic_path_desc\(ctypes\.Structure\):.*?\n(    _fields_ = \[.*?\]\n)\n", text, flags=re.S)
    assert m, "INTEGRATION.md no longer shows the nic_path_desc binding"
    ns = {"ctypes": ctypes}
    exec("class nic_path_desc(ctypes.Structure):\n" + m.group(1), ns)
    doc = ns["nic_path_desc"]
    from neural_image_compression_v2_amd._lib import NicPathDesc
    assert [f[0] for f in doc._fields_] == [f[0] for f in NicPathDesc._fields_]
    prog = ['#include <stdio.h>', f'#include "{HEADER}"', 'int main(){printf("%zu\\n", sizeof(nic_path_desc)); return 0;}']
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write("\n".join(prog))
        subprocess.run(["gcc", "-std=c11", src, "-o", exe], check=True)
        size = int(subprocess.run([exe], capture_output=True, text=True, check=True).stdout)
    assert ctypes.sizeof(doc) == size == ctypes.sizeof(NicPathDesc)
    for f, _ in doc._fields_:
        assert getattr(doc, f).offset == getattr(NicPathDesc, f).offset, f


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no rank environment must start two ranks itself (torch.distributed.run as a child process)
    and relay rank 0's JSON line; a rank whose WORLD_SIZE disagrees with --gpus exits non-zero.  --launch-check keeps the GPU out
    of it (gloo rendezvous on 127.0.0.1 only)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    import json
    rec = json.loads(lines[0])
    assert rec["launch_check"] is True and rec["n_gpus"] == 2 and rec["rank_sum"] == 1.0
    assert rec["stripes"] == [[0, 1920], [1920, 1920]] and rec["covered"] == 3840          # the 4K image's last sample axis, tiled by the two ranks
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env2, capture_output=True,
                        text=True, timeout=120)
    assert r2.returncode == 2 and "WORLD_SIZE=1 but --gpus 2" in r2.stderr


@pytest.mark.parametrize("extra,key,want", [
    (["--workload", "video", "--scaling", "strong"], "stripes", [[0, 640], [640, 640], [1280, 640]]),
    (["--workload", "video"], "covered", 1920),
    (["--workload", "fits64"], "fits_per_rank", [21, 21, 22]),
    (["--scaling", "strong", "--shard", "replicated"], "covered", 3840),
])
def test_bench_multi_gpu_workloads_launch(extra, key, want):
    """every BASELINE multi-GPU configuration has a runnable N-rank command: --workload video / fits64 with --gpus N and
    the strong-scaling modes of the headline; --launch-check keeps the GPU out of it and prints the per-rank plan"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"] + extra, env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 3 and rec["rank_sum"] == 3.0 and rec[key] == want, rec


def _video_stripe_worker(rank, world, port, out_path, overlapped=False):
    """3D field, z-stripes, STRONG scaling (one pass over the field split over the ranks), Adam over the rank's own node rows only;
    ``overlapped``: the interior rows are updated from inside ``stripe_exchange(overlap=)`` - before the sums are written back - and only the
    boundary rows afterwards (the ordering of the multi-GPU bench step)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from neural_image_compression_v2_amd import _lib, fused
        from neural_image_compression_v2_amd.distributed import (assemble_stripes, plan_stripes, stripe_exchange, stripe_param_blocks, stripe_row_parts,
                                                                     stripe_state)
        from oracle import nic_oracle as O
        torch.set_num_threads(2)
        ext = (8, 8, 32)                                             # sample axes (x, y, z); z is the stripe axis (tensor axis 1 of [C, Z, Y, X])
        g = torch.Generator().manual_seed(4)
        fp, _ = O.create_pyramid(tuple(e // 4 for e in ext), 12, 8, dim=3, no_mip=True, generator=g)     # [12, 9, 3, 3], [12, 5, 2, 2]
        g0, g1 = fp[0].detach().clone(), fp[1].detach().clone()
        mlp = O.init_mlp(79, 64, generator=g)
        field = torch.rand(*ext, 3, generator=g)
        plan = plan_stripes(ext[2], 8, rank, world)
        n_global = ext[0] * ext[1] * ext[2]                          # strong scaling: every voxel once per step over the whole job

        def geo_of(extent, base, noise=_lib.NIC_NOISE_KERNEL):
            return fused.PathGeometry(dim=3, method=4, step_number=0.25, mip_level=0, extent=extent, num_crops=1,
                                      noise_mode=noise, noise_seed=5, noise_offset=2, sample_base=base, loss_scale=1.0 / (3.0 * n_global))
        local = (ext[0], ext[1], plan.size)
        tgt = field[:, :, plan.start:plan.start + plan.size].reshape(-1, 3)
        out = _oracle_step(geo_of(local, ext[0] * ext[1] * plan.start), g0, g1, torch.tensor([[0, 0, plan.start]]), mlp.tensors(), tgt)
        offs, sizes, _ = fused.grad_bucket_layout(geo_of(local, 0), g0, g1)
        # "Adam" stand-in with state: p -= 0.5 * (grad + m), moments start NON-ZERO outside the stripe too - only the own rows may move
        moved = []
        levels = ((g0, out.grad_g0), (g1, out.grad_g1))
        befores = [p.clone() for p, _ in levels]
        if not overlapped:
            stripe_exchange(plan, out.flat[:offs[7]], out.grad_g0, out.grad_g1)
            for level, (p, gr) in enumerate(levels):
                m = stripe_state(plan, level, p) + 0.25              # a resumed state: non-zero moments
                for c, (pb, gb) in enumerate(stripe_param_blocks(plan, level, p, gr)):
                    assert pb.is_contiguous() and gb.is_contiguous() and pb.shape == m[c].shape
                    pb -= 0.5 * (gb + m[c])
        else:
            def rows_update(level, r0, r1):
                p, gr = levels[level]
                if r0 <= r1:
                    p[:, r0:r1 + 1] -= 0.5 * (gr[:, r0:r1 + 1] + 0.25)
            parts = [stripe_row_parts(plan, level) for level in (0, 1)]
            for level in (0, 1):                                     # the split tiles the own rows exactly
                (ilo, ihi), brows = parts[level]
                lo, hi = plan.node_rows(level)
                assert sorted(list(range(ilo, ihi + 1)) + brows) == list(range(lo, hi + 1))
                assert all(r in plan.boundary_rows(level) for r in brows) and not any(r in plan.boundary_rows(level) for r in range(ilo, ihi + 1))
            snap = [gr.clone() for _, gr in levels]

            def interior():
                # called with the collective in flight: the gradients of the interior rows are already final (nothing of the exchange is in yet)
                assert all(torch.equal(a, gr) for a, (_, gr) in zip(snap, levels))
                for level in (0, 1):
                    rows_update(level, *parts[level][0])
            stripe_exchange(plan, out.flat[:offs[7]], out.grad_g0, out.grad_g1, overlap=interior)
            for level in (0, 1):
                for r in parts[level][1]:
                    rows_update(level, r, r)                         # boundary rows: with the summed gradients
        for level, (p, _) in enumerate(levels):
            lo, hi = plan.node_rows(level)
            moved.append(float((p - befores[level])[:, :lo].abs().sum() + (p - befores[level])[:, hi + 1:].abs().sum()))
        assemble_stripes(plan, g0, g1)
        # single process: the same voxels as the ranks' stripes in rank order, global sample ids
        plans = [plan_stripes(ext[2], 8, r, world) for r in range(world)]
        singles = [_oracle_step(geo_of((ext[0], ext[1], q.size), ext[0] * ext[1] * q.start), fp[0].detach(), fp[1].detach(),
                                torch.tensor([[0, 0, q.start]]), mlp.tensors(), field[:, :, q.start:q.start + q.size].reshape(-1, 3)) for q in plans]
        sflat = sum(sg.flat for sg in singles)
        sg0, sg1 = sum(sg.grad_g0 for sg in singles), sum(sg.grad_g1 for sg in singles)
        # without noise (its draws are keyed by the sample id, and a stripe numbers its samples on its own extent) the stripes of one pass
        # ARE the whole-field pass: losses and gradients add up to the single whole-field step
        quiet = [_oracle_step(geo_of((ext[0], ext[1], q.size), 0, _lib.NIC_NOISE_NONE), fp[0].detach(), fp[1].detach(),
                              torch.tensor([[0, 0, q.start]]), mlp.tensors(), field[:, :, q.start:q.start + q.size].reshape(-1, 3)) for q in plans]
        sflat_q, sg0_q = sum(sg.flat for sg in quiet), sum(sg.grad_g0 for sg in quiet)
        whole = _oracle_step(geo_of(ext, 0, _lib.NIC_NOISE_NONE), fp[0].detach(), fp[1].detach(), torch.tensor([[0, 0, 0]]), mlp.tensors(), field.reshape(-1, 3))
        res = {"small": float((out.flat[:offs[7]] - sflat[:offs[7]]).abs().max() / sflat[:offs[7]].abs().max()),
               "moved_outside": max(moved),
               "p0": float((g0 - (fp[0].detach() - 0.5 * (sg0 + 0.25))).abs().max()), "p1": float((g1 - (fp[1].detach() - 0.5 * (sg1 + 0.25))).abs().max()),
               # the stripes' sample ids tile the single whole-field pass: same loss, same gradients (the noise is keyed by the global id)
               "whole_loss": float((sflat_q[0] - whole.flat[0]).abs() / whole.flat[0].abs()), "whole_g0": float((sg0_q - whole.grad_g0).abs().max() / whole.grad_g0.abs().max())}
        torch.save(res, out_path + f".{rank}")
    finally:
        dist.destroy_process_group()


def test_video_stripes_strong_scaling_and_stripe_owned_adam_gloo():
    """BASELINE config 4's shape at reduced size over 2 gloo ranks: z-stripes of a 3D field (method 4), STRONG scaling - the stripes of one
    whole-field pass -, the small exchange, an optimiser update restricted to the rank's own node rows (non-zero moments elsewhere must not
    move anything: the resume case round 2 documented as broken), assembled grids == the single-process update; and - noise off - the stripes'
    losses / gradients add up to the whole-field single-process step."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r.pt")
        mp.spawn(_video_stripe_worker, args=(2, port, out), nprocs=2, join=True)
        for r in range(2):
            res = torch.load(out + f".{r}")
            assert res["moved_outside"] == 0.0, res
            assert res["small"] < 1e-6 and res["p0"] < 1e-6 and res["p1"] < 1e-6, (r, res)
            assert res["whole_loss"] < 1e-6 and res["whole_g0"] < 1e-5, (r, res)


def test_overlapped_stripe_step_matches_single_process_gloo():
    """the overlapped ordering of the stripe-sharded step - exchange started, the optimiser over the INTERIOR node rows
    while it is in flight (their gradients must already be final: the callback sees the un-exchanged bucket), the boundary rows after it - gives
    the single-process update on every rank; the interior / boundary split tiles a rank's own rows exactly"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r.pt")
        mp.spawn(_video_stripe_worker, args=(2, port, out, True), nprocs=2, join=True)
        for r in range(2):
            res = torch.load(out + f".{r}")
            assert res["moved_outside"] == 0.0, res
            assert res["small"] < 1e-6 and res["p0"] < 1e-6 and res["p1"] < 1e-6, (r, res)


def test_counter_based_sampler_host_functions_match_the_oracle(lib):
    """the library's host-side sampler functions (the LOD is always drawn on the host; the origins function is the
    host twin of the device kernel) against the oracle's restatement of the generator, and the laws the reference's draws have:
    P(lod = k) = 3/4 4^-k clamped at MAX_MIP (image_compression.py:32-34), uniform LODs on 0..MAX_MIP (:30), origins uniform on
    [0, data_size - crop + 1) (:40-41)."""
    from oracle import nic_oracle as O
    for seed in (0, 7, 0x123456789ABCDEF):
        for step in (0, 1, 5, 1000, 2 ** 33 + 5):
            for uni in (0, 1):
                for mm in (0, 3, 9):
                    assert lib.nic_sampler_lod_host(seed, step, uni, mm) == O.sampler_lod(seed, step, bool(uni), mm)
            for dim, rng in ((2, 257), (3, 33), (2, 1)):
                arr = (ctypes.c_int32 * (8 * dim))()
                assert lib.nic_sampler_origins_host(seed, step, 8, dim, rng, arr) == 0
                assert list(arr) == O.sampler_origins(seed, step, 8, dim, rng).reshape(-1).tolist()
                assert min(arr) >= 0 and max(arr) < rng
    n = 20000
    lods = np.array([lib.nic_sampler_lod_host(3, s, 0, 9) for s in range(n)])
    for k in range(4):
        assert abs((lods == k).mean() - 0.75 * 4.0 ** -k) < 0.01
    assert (np.array([lib.nic_sampler_lod_host(3, s, 0, 1) for s in range(2000)]) <= 1).all()
    uni = np.array([lib.nic_sampler_lod_host(3, s, 1, 4) for s in range(n)])
    assert all(abs((uni == k).mean() - 0.2) < 0.015 for k in range(5))
    arr = (ctypes.c_int32 * 2)()
    xs = []
    for s in range(4000):
        lib.nic_sampler_origins_host(11, s, 1, 2, 100, arr)
        xs.append((arr[0], arr[1]))
    xs = np.array(xs)
    assert abs(xs.mean() - 49.5) < 1.5 and abs(np.corrcoef(xs[:, 0], xs[:, 1])[0, 1]) < 0.05
    assert lib.nic_sampler_lod_host(0, 0, 0, -1) < 0 and lib.nic_sampler_origins_host(0, 0, 1, 2, 0, arr) < 0


def test_settings_resolve_the_product_precision():
    """TF_PLAIN_BF16 = 0 (default): the reference-faithful arithmetic (split products = fp32-equivalent) in every dimension - plain 16-bit products are an
    opt-in; -1 keeps round 3's "3D only" resolution; explicit values win; the argv form of the reference's flag system parses it"""
    from neural_image_compression_v2_amd.var2 import Settings
    assert Settings().plain_bf16 is False
    assert Settings(IMAGE_DIMENSION=3, COMPRESSION_METHOD=3).plain_bf16 is False and Settings(IMAGE_DIMENSION=3, COMPRESSION_METHOD=4).plain_bf16 is False
    assert Settings(IMAGE_DIMENSION=3, COMPRESSION_METHOD=3, TF_PLAIN_BF16=-1).plain_bf16 is True and Settings(TF_PLAIN_BF16=-1).plain_bf16 is False
    assert Settings(IMAGE_DIMENSION=3, COMPRESSION_METHOD=2, TF_PLAIN_BF16=-1).plain_bf16 is False   # method 2: a 3D volume flattened to a 2D pyramid
    assert Settings(IMAGE_DIMENSION=3, COMPRESSION_METHOD=3, TF_PLAIN_BF16=True).plain_bf16 is True and Settings(TF_PLAIN_BF16=True).plain_bf16 is True
    assert Settings.from_argv(["TF_PLAIN_BF16=1"]).plain_bf16 is True and Settings.from_argv(["IMAGE_DIMENSION=3", "TF_PLAIN_BF16=0"]).plain_bf16 is False
    assert Settings().TF_PLAIN_FP16 is False and Settings.from_argv(["TF_PLAIN_FP16=True"]).TF_PLAIN_FP16 is True


def test_light_cosine_scheduler_is_torchs_bit_for_bit():
    """optim.CosineAnnealing replaces torch's CosineAnnealingLR in the training loop (70 us of host time per step): the learning rates
    of both groups must be the same doubles over whole schedules, including steps past T_max (the restart branch)."""
    import torch
    from neural_image_compression_v2_amd.optim import CosineAnnealing
    for T in (1, 7, 100, 1000):
        ps = [torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))]
        mk = lambda: torch.optim.SGD([{"params": [ps[0]], "lr": 0.01}, {"params": [ps[1]], "lr": 0.005}])
        o_ref, o_new = mk(), mk()
        ref = torch.optim.lr_scheduler.CosineAnnealingLR(o_ref, T_max=T, eta_min=0)
        new = CosineAnnealing(o_new, T_max=T, eta_min=0)
        for step in range(2 * T + 5):
            o_ref.step(); ref.step(); new.step()
            assert ref.get_last_lr() == new.get_last_lr(), (T, step, ref.get_last_lr(), new.get_last_lr())
            assert [g["lr"] for g in o_ref.param_groups] == [g["lr"] for g in o_new.param_groups]
