#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE itself (container only) and stores inputs + outputs.

TEST INFRASTRUCTURE.  This script is the only place in the repository that touches
``/root/reference``.  It imports the reference's own library files (``Projects/fp_def.py``,
``utils.py``, ``models.py``, ``positional_encoding.py``) from where they lie and, because
``Projects/image_compression.py`` is a script with import-time side effects (data loading,
tensorboard, file creation), it loads only the *function/class definitions* this path needs out of
that file with ``ast`` at run time and executes them in a namespace that provides the configuration
globals they read.  No reference source text is written into this repository: the fixtures under
``tests/golden/`` hold numbers only (inputs and the reference's outputs).

The reference is never imported on the GPU box (it does not exist there); the committed ``.npz``
files travel instead.

Notes
* ``utils.py:2`` imports ``cv2`` which is absent from this image; none of the functions on the hot
  path touch it, so an empty in-memory module object is registered under that name before import
  (SURVEY.md section 8c).  Nothing is written to disk for it.
* bytecode writing is disabled so the read-only reference tree stays untouched.

Run:  python oracle/make_golden.py            (from the repository root, in the build container)
"""
from __future__ import annotations

import ast
import math
import os
import random
import sys
import tempfile
import types

sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference/Projects"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference_modules():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present: this script only runs in the build container")
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import models as ref_models  # noqa
    import utils as ref_utils  # noqa
    import fp_def as ref_fp  # noqa
    import positional_encoding as ref_pe  # noqa
    return ref_models, ref_utils, ref_fp, ref_pe


WANTED = {
    "random_crop_dataset", "ColorDecoder",
    "create_decoder_input_2d", "create_decoder_input_3d", "create_decoder_input_3d_v2",
    "finally_decode_input_2d", "finally_decode_input_3d", "finally_decode_input_3d_v2",
    "train_models", "decode_image",
}


def load_driver_defs(ns: dict):
    """Execute the wanted def/class nodes of image_compression.py inside ``ns``."""
    path = os.path.join(REF, "image_compression.py")
    with open(path, "r", encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    body = [n for n in tree.body
            if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in WANTED]
    missing = WANTED - {n.name for n in body}
    assert not missing, missing
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, path, "exec"), ns)


class _NullWriter:
    """sink for the SummaryWriter.add_scalar calls of train_models (logging only)."""

    def __init__(self):
        self.scalars = {}

    def add_scalar(self, tag, value, step):
        self.scalars.setdefault(tag, []).append((int(step), float(value)))


def driver_namespace(ref_models, ref_utils, ref_fp, **cfg):
    """Namespace with what ``from utils/models/fp_def/var2 import *`` would have provided."""
    ns = {}
    for m in (ref_utils, ref_models, ref_fp):
        ns.update({k: v for k, v in vars(m).items() if not k.startswith("__")})
    import torch.nn as nn
    import time
    ns.update(dict(torch=torch, nn=nn, np=np, math=math, random=random, time=time))
    d = dict(
        DEVICE=torch.device("cpu"), FP_BITS=8, NUM_EPOCHS=10, IMAGE_SIZE=256, MAX_MIP_LEVEL=0,
        FEATURE_PYRAMID_CHANNELS=12, PE_CHANNELS=6, COMPRESSION_METHOD=1, MLP_NUM_DTYPE=32,
        UNIFORM_DISTRIBUTION_RATE=0.05, IMAGE_DIMENSION=2, OUTPUT_BITS=8, HIDDEN_LAYER_CHANNELS=64,
        CROP_MIP_LEVEL=8, NUM_CROPS=8, INTERVAL_PRINT=10 ** 9, INTERVAL_SAVE_MODEL=10 ** 9,
        TF_NO_MIP=True, TF_USE_TRI_PE=True, TF_PRINT_LOG=False, TF_PRINT_PSNR=False,
        TF_WRITE_TIME=False, TF_WRITE_PSNR=False, PRINTLOG_PATH=os.devnull,
    )
    d.update(cfg)
    # derived values, following var2.py:188-202
    d["FEATURE_PYRAMID_SIZE"] = d["IMAGE_SIZE"] // 4
    d["FP_DIMENSION"] = 2 if d["COMPRESSION_METHOD"] == 2 else d["IMAGE_DIMENSION"]
    if d["TF_NO_MIP"]:
        d["MAX_MIP_LEVEL"] = 0
    C, P, D = d["FEATURE_PYRAMID_CHANNELS"], d["PE_CHANNELS"], d["FP_DIMENSION"]
    d["DECODER_INPUT_CHANNELS"] = C * (2 ** D + 1) + P * D + 1
    if d["COMPRESSION_METHOD"] == 4:
        d["DECODER_INPUT_CHANNELS"] = C * (2 ** 2 + 1) + P * D + 1
    d["CROP_SIZE"] = 2 ** d["CROP_MIP_LEVEL"]
    d["MLP_DTYPE"] = ref_utils.bits2dtype_torch(d["MLP_NUM_DTYPE"], "float")
    ns.update(d)
    load_driver_defs(ns)
    return ns


def T(x):
    """tensor / scalar -> numpy for storage"""
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: T(v) for k, v in arrays.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(arrays)} arrays)")


def digest(x: torch.Tensor):
    """size-independent fingerprint of a large tensor: [sum, sum of squares, dot with a fixed
    deterministic probe vector], all in float64."""
    v = x.detach().reshape(-1).to(torch.float64)
    idx = torch.arange(v.numel(), dtype=torch.float64)
    probe = torch.frac(torch.sin(idx * 12.9898 + 0.5) * 43758.5453)
    return np.array([v.sum().item(), (v * v).sum().item(), (v * probe).sum().item()])


# ----------------------------------------------------------------------------------------------
def g1_levels(ref_fp):
    out = {}
    for image_size, base in [(512, 128), (64, 16), (1024, 256), (256, 64), (128, 32), (16, 4)]:
        d = ref_fp.create_pyramid_mip_levels(image_size, base)
        keys = sorted(d.keys())
        out[f"map_{image_size}_{base}"] = np.array([[k, d[k]] for k in keys], dtype=np.int64)
    sizes = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 4096]
    out["sizes"] = np.array(sizes)
    out["two_power"] = np.array([ref_fp.return_2_power(s) for s in sizes])
    out["levels"] = np.array([ref_fp.return_pyramid_levels(s) for s in sizes])
    save("levels", **out)


def g2_pe(ref_utils, ref_pe):
    out = {}
    dt = torch.float32
    c1 = torch.tensor([[0, .125, .25, .5, 1, 1.5, 2, 3.875]], dtype=dt)
    out["tri_c1_in"] = c1
    out["tri_c1"] = ref_utils.triangular_positional_encoding(c1, 6, "cpu", dt)
    t14 = torch.tensor([[0, 1, 2, 3, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1, 2, 3]])   # test14.py:96
    out["tri_test14_in"] = t14
    out["tri_test14"] = ref_utils.triangular_positional_encoding(t14, 6, "cpu", dt)
    g = torch.Generator().manual_seed(11)
    for D in (2, 3):
        frac = torch.randint(0, 480 * 8, (D, 257), generator=g).to(dt) / 8          # multiples of 1/8
        out[f"coords_d{D}"] = frac
        out[f"tri_d{D}"] = ref_utils.triangular_positional_encoding(frac, 6, "cpu", dt)
        out[f"sin_d{D}"] = ref_utils.positional_encoding(tuple(frac[i] for i in range(D)), 6, "cpu", dt)
        out[f"tri_d{D}_p4"] = ref_utils.triangular_positional_encoding(frac, 4, "cpu", dt)
        out[f"sin_d{D}_p8"] = ref_utils.positional_encoding(tuple(frac[i] for i in range(D)), 8, "cpu", dt)
    arb = (torch.rand(2, 100, generator=g) * 300 - 20).to(dt)                          # arbitrary reals, negative too
    out["coords_arb"] = arb
    out["tri_arb"] = ref_utils.triangular_positional_encoding(arb, 6, "cpu", dt)
    out["sin_arb"] = ref_utils.positional_encoding((arb[0], arb[1]), 6, "cpu", dt)
    # G3: nn.Module LUT form and its functional twins
    m = ref_pe.TriangularPositionalEncoding1D()
    out["lut1d_encodings"] = m.encodings
    q = torch.tensor([[0, 1, 2, 3, 9], [7, 8, 15, 16, 100]])
    out["lut1d_in"] = q
    out["lut1d_out"] = m(q)
    m2 = ref_pe.TriangularPositionalEncoding1D(sequence_length=16, octaves=4, include_constant=False)
    out["lut1d_16_4_encodings"] = m2.encodings
    out["lut1d_16_4_out"] = m2(q)
    out["fn1d"] = ref_utils.triangular_positional_encoding_1d("cpu", dt)
    out["fn2d_00_8_8"] = ref_utils.triangular_positional_encoding_2d(torch.tensor([[0, 0]]), 8, 8, "cpu", dt)   # test14.py:88-92
    cc = torch.tensor([[3, 5], [10, 2]])
    out["fn2d_in"] = cc
    out["fn2d_4_4"] = ref_utils.triangular_positional_encoding_2d(cc, 4, 4, "cpu", dt)
    out["fn2d_4_4_s2"] = ref_utils.triangular_positional_encoding_2d(cc, 4, 4, "cpu", dt, stride=2)
    fx, fy = ref_utils.convert_coordinate_start(cc, 4, 4, "cpu", dt)
    out["ccs_x"], out["ccs_y"] = fx, fy
    save("pe", **out)


def g4_g0g1_2d(ref_fp):
    out = {}
    dt = torch.float32
    torch.manual_seed(21)
    fp, levels = ref_fp.create_pyramid(16, 3, 8, "cpu", dt, True)           # [3,17,17],[3,9,9]
    assert levels == 1
    for i, g in enumerate(fp):
        out[f"nomip_grid{i}"] = g
    names = ["g0_0", "g0_1", "g0_2", "g0_3", "g1_0", "g1_1", "g1_2", "g1_3", "pe"]
    for tag, tri in (("tri", True), ("sin", False)):
        res = ref_fp.create_g0_g1(fp, 0, torch.tensor(3), torch.tensor(5), 0.25,
                                  torch.arange(8), torch.arange(8), 6, "cpu", dt, tri)
        for n, r in zip(names, res):
            out[f"nomip_{tag}_{n}"] = r
    # rectangular ranges (x_range != y_range length)
    res = ref_fp.create_g0_g1(fp, 0, torch.tensor(1), torch.tensor(50), 0.25,
                              torch.arange(12), torch.arange(5), 6, "cpu", dt, True)
    for n, r in zip(names, res):
        out[f"rect_{n}"] = r
    # mip pyramid: every step_number the reference can produce, incl. the unweighted step==2 case (Q6)
    torch.manual_seed(22)
    fpm, levels = ref_fp.create_pyramid(16, 3, 8, "cpu", dt, False)          # 17,9,5,3
    assert levels == 2
    for i, g in enumerate(fpm):
        out[f"mip_grid{i}"] = g
    cases = []
    for fl, mip, S, ox, oy in [(0, 0, 8, 3, 5), (0, 1, 8, 2, 7), (0, 2, 8, 1, 4), (0, 3, 4, 3, 1),
                               (1, 4, 2, 1, 0), (1, 5, 1, 1, 1), (1, 6, 1, 0, 0)]:
        step = pow(2, mip - (fl + 1) * 2)                                     # image_compression.py:79
        res = ref_fp.create_g0_g1(fpm, fl, torch.tensor(ox), torch.tensor(oy), step,
                                  torch.arange(S), torch.arange(S), 6, "cpu", dt, True)
        k = len(cases)
        cases.append([fl, mip, S, ox, oy])
        for n, r in zip(names, res):
            out[f"mip_case{k}_{n}"] = r
    out["mip_cases"] = np.array(cases, dtype=np.int64)
    save("g0g1_2d", **out)


def g5_g0g1_3d(ref_fp):
    out = {}
    dt = torch.float32
    torch.manual_seed(31)
    fp, _ = ref_fp.create_pyramid_3d(16, 2, 8, "cpu", dt, True)              # [2,17,17,17],[2,9,9,9]
    for i, g in enumerate(fp):
        out[f"grid{i}"] = g
    S = 8
    rng = torch.arange(S, dtype=dt)                                           # image_compression.py:116-118
    r3 = ref_fp.create_g0_g1_3d(fp, 0, torch.tensor(3), torch.tensor(5), torch.tensor(9), 0.25,
                                rng, rng, rng, 6, "cpu", dt)
    n3 = [f"g0_{i}" for i in range(8)] + [f"g1_{i}" for i in range(8)] + ["pe"]
    for n, r in zip(n3, r3):
        out[f"m3_{n}"] = r
    r4 = ref_fp.create_g0_g1_3d_v2(fp, 0, torch.tensor(3), torch.tensor(5), torch.tensor(9), 0.25,
                                   rng, rng, rng, 6, "cpu", dt)
    n4 = [f"g0_{i}" for i in range(4)] + [f"g1_{i}" for i in range(8)] + ["pe"]
    for n, r in zip(n4, r4):
        out[f"m4_{n}"] = r
    # mip pyramid in 3D, step 1/2 (weights on) and step 2 (Q6: unweighted)
    torch.manual_seed(32)
    fpm, levels = ref_fp.create_pyramid_3d(16, 2, 8, "cpu", dt, False)
    for i, g in enumerate(fpm):
        out[f"mip_grid{i}"] = g
    cases = []
    for fl, mip, S, o in [(0, 1, 4, (2, 7, 1)), (0, 3, 2, (3, 1, 0)), (1, 4, 2, (1, 0, 1))]:
        step = pow(2, mip - (fl + 1) * 2)
        rg = torch.arange(S, dtype=dt)
        for tag, fn, nn_ in (("m3", ref_fp.create_g0_g1_3d, n3), ("m4", ref_fp.create_g0_g1_3d_v2, n4)):
            res = fn(fpm, fl, torch.tensor(o[0]), torch.tensor(o[1]), torch.tensor(o[2]), step, rg, rg, rg, 6, "cpu", dt)
            for n, r in zip(nn_, res):
                out[f"mip_case{len(cases)}_{tag}_{n}"] = r
        cases.append([fl, mip, S, *o])
    out["mip_cases"] = np.array(cases, dtype=np.int64)
    save("g0g1_3d", **out)


def g6_decoder_input(ref_models, ref_utils, ref_fp):
    out = {}
    dt = torch.float32
    # ---- 2D, mip pyramid, mip 4 (step 1) and mip 5 (step 2 -> unweighted G1), C = 3
    ns = driver_namespace(ref_models, ref_utils, ref_fp, IMAGE_SIZE=256, FEATURE_PYRAMID_CHANNELS=3,
                          TF_NO_MIP=False, MAX_MIP_LEVEL=8)
    torch.manual_seed(41)
    fp, levels = ref_fp.create_pyramid(ns["FEATURE_PYRAMID_SIZE"], 3, 8, "cpu", dt, False)
    for i, g in enumerate(fp):
        out[f"d2_grid{i}"] = g
    mp = ref_fp.create_pyramid_mip_levels(256, 64)
    ns["feature_pyramid_mip_levels_dict"] = mp
    for mip in (4, 5):
        fl = mp[mip]
        coord = torch.zeros(2, 2, dtype=torch.int64)          # crop == whole mip image (origin range is [0,1))
        for tri in (True, False):
            ns["TF_USE_TRI_PE"] = tri
            x = ns["create_decoder_input_2d"](fp, coord, 2, fl, mip)
            out[f"d2_mip{mip}_{'tri' if tri else 'sin'}"] = x
    ns["TF_USE_TRI_PE"] = True
    # finally_decode_input_2d on a tile at a non-zero origin (mip 2: 64x64 image, tile 16 at (16, 32))
    out["d2_final_mip2_tile"] = ns["finally_decode_input_2d"](fp, 16, 2, 16, 32)
    out["d2_final_mip3_full"] = ns["finally_decode_input_2d"](fp, 32, 3)
    # ---- 2D, mip 0, the default training shape (256 x 256 samples per crop): digest + row sample
    ns0 = driver_namespace(ref_models, ref_utils, ref_fp, IMAGE_SIZE=512, FEATURE_PYRAMID_CHANNELS=3)
    torch.manual_seed(42)
    fp0, _ = ref_fp.create_pyramid(ns0["FEATURE_PYRAMID_SIZE"], 3, 8, "cpu", dt, True)   # [3,129,129],[3,65,65]
    out["d2m0_grid0"], out["d2m0_grid1"] = fp0
    coord = torch.tensor([[17, 201], [256, 0]])
    x = ns0["create_decoder_input_2d"](fp0, coord, 2, 0, 0)
    out["d2m0_coord"] = coord
    out["d2m0_shape"] = np.array(x.shape)
    out["d2m0_digest"] = digest(x)
    rows = torch.arange(0, x.shape[0], 997)
    out["d2m0_rows"] = rows
    out["d2m0_sample"] = x[rows]
    # ---- 3D method 3 and method 4, CROP_MIP_LEVEL = 3 (8^3 samples per crop), C = 2
    for method, fname in ((3, "create_decoder_input_3d"), (4, "create_decoder_input_3d_v2")):
        ns3 = driver_namespace(ref_models, ref_utils, ref_fp, IMAGE_SIZE=64, IMAGE_DIMENSION=3,
                               COMPRESSION_METHOD=method, FEATURE_PYRAMID_CHANNELS=2, CROP_MIP_LEVEL=3)
        torch.manual_seed(43)
        fp3, _ = ref_fp.create_pyramid_3d(16, 2, 8, "cpu", dt, True)
        if method == 3:
            out["d3_grid0"], out["d3_grid1"] = fp3
        coord = torch.tensor([[3, 5, 9], [56, 0, 31]])
        out["d3_coord"] = coord
        out[f"d3_m{method}"] = ns3[fname](fp3, coord, 2, 0, 0)
        ns3["feature_pyramid_mip_levels_dict"] = ref_fp.create_pyramid_mip_levels(64, 16)
        fin = "finally_decode_input_3d" if method == 3 else "finally_decode_input_3d_v2"
        out[f"d3_final_m{method}"] = ns3[fin](fp3, 4, 0, 8, 20, 60)
    save("decoder_input", **out)


def g7_g8_mlp_fwdbwd(ref_models, ref_utils, ref_fp):
    """ColorDecoder forward, and the full noisy forward + MSE + backward of one training step,
    composed exactly as train_models does it (image_compression.py:239-265)."""
    out = {}
    dt = torch.float32
    cases = [
        # tag, cfg, builder name, pyramid fn, coords, fl, mip
        ("d2", dict(IMAGE_SIZE=256, FEATURE_PYRAMID_CHANNELS=12, TF_NO_MIP=False, MAX_MIP_LEVEL=8),
         "create_decoder_input_2d", "create_pyramid", torch.zeros(2, 2, dtype=torch.int64), None, 4),
        ("d3m3", dict(IMAGE_SIZE=32, IMAGE_DIMENSION=3, COMPRESSION_METHOD=3, CROP_MIP_LEVEL=2),
         "create_decoder_input_3d", "create_pyramid_3d", torch.tensor([[3, 5, 9], [28, 0, 17]]), 0, 0),
        ("d3m4", dict(IMAGE_SIZE=32, IMAGE_DIMENSION=3, COMPRESSION_METHOD=4, CROP_MIP_LEVEL=2),
         "create_decoder_input_3d_v2", "create_pyramid_3d", torch.tensor([[3, 5, 9], [28, 0, 17]]), 0, 0),
    ]
    for tag, cfg, builder, pyr, coord, fl, mip in cases:
        ns = driver_namespace(ref_models, ref_utils, ref_fp, **cfg)
        torch.manual_seed(51)
        fp, _ = getattr(ref_fp, pyr)(ns["FEATURE_PYRAMID_SIZE"], ns["FEATURE_PYRAMID_CHANNELS"], 8, "cpu", dt, ns["TF_NO_MIP"])
        if fl is None:
            fl = ref_fp.create_pyramid_mip_levels(ns["IMAGE_SIZE"], ns["FEATURE_PYRAMID_SIZE"])[mip]
        decoder = ns["ColorDecoder"]()
        sd = decoder.state_dict()
        for k, v in sd.items():
            out[f"{tag}_sd_{k}"] = v.clone()
        out[f"{tag}_grid_g0"] = fp[2 * fl].detach().clone()
        out[f"{tag}_grid_g1"] = fp[2 * fl + 1].detach().clone()
        out[f"{tag}_coord"] = coord
        out[f"{tag}_fl_mip"] = np.array([fl, mip])
        x = ns[builder](fp, coord, coord.shape[0], fl, mip)
        noise = (torch.rand_like(x) - 0.5) / (2 ** 8)                      # image_compression.py:250
        target = torch.rand(x.shape[0], 3)
        y_clean = decoder(x)
        y = decoder(x + noise)
        loss = torch.nn.MSELoss()(y, target)
        loss.backward()
        out[f"{tag}_x"] = x
        out[f"{tag}_noise"] = noise
        out[f"{tag}_target"] = target
        out[f"{tag}_y_clean"] = y_clean
        out[f"{tag}_y"] = y
        out[f"{tag}_loss"] = loss
        out[f"{tag}_grad_g0"] = fp[2 * fl].grad
        out[f"{tag}_grad_g1"] = fp[2 * fl + 1].grad
        for k, p in decoder.named_parameters():
            out[f"{tag}_grad_{k}"] = p.grad
    save("fwdbwd", **out)


def g8b_fwdbwd_mip0(ref_models, ref_utils, ref_fp):
    """The default training shape (2D, no-mip, C = 12, two 256 x 256 crops, tri PE and sin PE).  The grids and the decoder
    (drawn by the reference from the torch generator) are STORED; noise and targets come from the oracle's counter-based
    generator (integer arithmetic in numpy: the same numbers on any torch version), so the test can rebuild every input
    without depending on the torch RNG stream; outputs are pinned by digests, row samples and the full set of decoder gradients."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import nic_oracle as O
    out = {}
    dt = torch.float32
    for tag, tri in (("tri", True), ("sin", False)):
        ns = driver_namespace(ref_models, ref_utils, ref_fp, IMAGE_SIZE=256, TF_USE_TRI_PE=tri)
        seed = 58 if tri else 59
        out[f"{tag}_seed"] = np.array(seed)
        torch.manual_seed(seed)
        fp, _ = ref_fp.create_pyramid(ns["FEATURE_PYRAMID_SIZE"], 12, 8, "cpu", dt, True)      # [12,65,65],[12,33,33]
        decoder = ns["ColorDecoder"]()
        coord = torch.zeros(2, 2, dtype=torch.int64)
        x = ns["create_decoder_input_2d"](fp, coord, 2, 0, 0)
        noise = O.kernel_noise(x.shape[0], 73, 8, seed=seed, offset=1)                   # portable: Threefry in numpy integers
        target = (O.kernel_noise(x.shape[0], 73, 0, seed=seed, offset=2)[:, :3] + 0.5).contiguous()      # U on a 2^-6 lattice of [0, 1)
        out[f"{tag}_g0"] = fp[0].detach().clone()
        out[f"{tag}_g1"] = fp[1].detach().clone()
        for k, p in decoder.state_dict().items():
            out[f"{tag}_sd_{k}"] = p.detach().clone()
        y = decoder(x + noise)
        loss = torch.nn.MSELoss()(y, target)
        loss.backward()
        out[f"{tag}_grid_digest"] = np.stack([digest(fp[0]), digest(fp[1])])
        out[f"{tag}_x_digest"] = digest(x)
        out[f"{tag}_noise_digest"] = digest(noise)
        out[f"{tag}_target_digest"] = digest(target)
        rows = torch.arange(0, x.shape[0], 1009)
        out[f"{tag}_rows"] = rows
        out[f"{tag}_x_rows"] = x[rows]
        out[f"{tag}_y_rows"] = y[rows]
        out[f"{tag}_y_digest"] = digest(y)
        out[f"{tag}_loss"] = loss
        out[f"{tag}_grad_g0_digest"] = digest(fp[0].grad)
        out[f"{tag}_grad_g1_digest"] = digest(fp[1].grad)
        out[f"{tag}_grad_g0_c0"] = fp[0].grad[0]              # one full channel plane of each grid gradient
        out[f"{tag}_grad_g1_c11"] = fp[1].grad[11]
        for k, p in decoder.named_parameters():
            out[f"{tag}_grad_{k}"] = p.grad
    save("fwdbwd_mip0", **out)


def g9_codec(ref_models, ref_utils):
    out = {}
    t = torch.tensor([-.5, -.25, 0, .002, .25, .499, .5])
    out["kat_in"] = t
    out["kat_save8"] = ref_models.save4fp(t, 8, torch.uint8)
    out["kat_load8"] = ref_models.load4fp(ref_models.save4fp(t, 8, torch.uint8), 8, torch.float32)
    out["kat_q4fp8"] = ref_models.quantize4fp(t, 8)
    for b in (2, 4, 8):
        q_min = -(pow(2, b) - 1) / pow(2, b + 1)
        a = torch.linspace(q_min, 0.5, steps=64)                               # test12.py:10
        out[f"ladder{b}_in"] = a
        out[f"ladder{b}_q4fp"] = ref_models.quantize4fp(a, b)
        out[f"ladder{b}_save"] = ref_models.save4fp(a, b, torch.uint8)
        out[f"ladder{b}_load"] = ref_models.load4fp(ref_models.save4fp(a, b, torch.uint8), b, torch.float32)
        out[f"ladder{b}_clamp"] = ref_models.quantize_clamp(a * 1.5, b)
    g = torch.Generator().manual_seed(61)
    u = torch.rand(1000, generator=g)
    out["u"] = u
    out["u_quantize8"] = ref_models.quantize(u, 8)
    out["u_to_bit8"] = ref_models.quantize_to_bit(u, 8)
    out["u_np_to_bit8"] = ref_models.quantize_to_bit(u.numpy(), 8)
    out["u_from_bit_to_bit"] = ref_models.quantize_from_bit_to_bit(u.numpy() * 255, 8)
    a = torch.tensor([0., 10.]); b = torch.tensor([1., 10.])
    out["psnr_kat"] = ref_utils.calculate_psnr(a, b)                           # 51.1751 dB
    v = torch.rand(64, 3, generator=g); w = torch.rand(64, 3, generator=g)
    out["psnr_a"], out["psnr_b"] = v, w
    out["psnr_torch"] = ref_utils.calculate_psnr(ref_models.quantize_to_bit(v, 8), ref_models.quantize_to_bit(w, 8))
    out["psnr_np"] = ref_utils.calculate_psnr(v.numpy() * 255, w.numpy() * 255)
    out["psnr_bits10"] = ref_utils.calculate_psnr(v, w, 10)
    # fp_* list helpers on a tiny pyramid
    import fp_def as ref_fp
    torch.manual_seed(62)
    fp, _ = ref_fp.create_pyramid(8, 2, 4, "cpu", torch.float32, False)
    for i, gq in enumerate(fp):
        out[f"fp_grid{i}"] = gq
    sav = ref_fp.fp_savable(fp, 4, torch.uint8)
    for i, gq in enumerate(sav):
        out[f"fp_sav{i}"] = gq
    lod = ref_fp.fp_load(sav, 4, torch.float32)
    for i, gq in enumerate(lod):
        out[f"fp_load{i}"] = gq
    allq = ref_fp.fp_all_quantize(fp, 4)
    for i, gq in enumerate(allq):
        out[f"fp_allq{i}"] = gq
    big = [(g_.detach() * 1.3).clone() for g_ in fp]
    ref_fp.fp_quantize_clamp(big, 1, 4)
    for i, gq in enumerate(big):
        out[f"fp_clamp1_{i}"] = gq
    out["q_range_bits"] = np.array([2, 4, 8])
    out["q_min"] = np.array([-(pow(2, b) - 1) / pow(2, b + 1) for b in (2, 4, 8)])
    save("codec", **out)


def g10_trajectory(ref_models, ref_utils, ref_fp):
    """Runs the reference's own train_models (image_compression.py:215-303) for a few steps on a
    synthetic image and records per-step (lod, crop origins, loss) and the final parameters.

    The noise and the crop origins come from torch's global CPU generator and python's ``random``;
    the fixture stores the seeds, and the oracle's replay draws the same streams in the same order
    (same torch build on the GPU box)."""
    out = {}
    dt = torch.float32
    runs = [
        ("d3m3", dict(IMAGE_SIZE=16, IMAGE_DIMENSION=3, COMPRESSION_METHOD=3, CROP_MIP_LEVEL=2, NUM_CROPS=2,
                      FEATURE_PYRAMID_CHANNELS=4, NUM_EPOCHS=40)),
        ("d3m4", dict(IMAGE_SIZE=16, IMAGE_DIMENSION=3, COMPRESSION_METHOD=4, CROP_MIP_LEVEL=2, NUM_CROPS=2,
                      FEATURE_PYRAMID_CHANNELS=4, NUM_EPOCHS=40)),
        ("d2mip", dict(IMAGE_SIZE=256, IMAGE_DIMENSION=2, COMPRESSION_METHOD=1, NUM_CROPS=2, TF_NO_MIP=False,
                       MAX_MIP_LEVEL=8, FEATURE_PYRAMID_CHANNELS=4, NUM_EPOCHS=24, UNIFORM_DISTRIBUTION_RATE=0.3,
                       TF_USE_TRI_PE=False)),
    ]
    for tag, cfg in runs:
        ns = driver_namespace(ref_models, ref_utils, ref_fp, **cfg)
        D = ns["FP_DIMENSION"]
        seed = 70 + len(out)
        out[f"{tag}_seed"] = np.array(seed)
        torch.manual_seed(seed)
        random.seed(seed)
        # synthetic image pyramid (what image_compression.py:429-477 would have produced: one tensor
        # [3, S, S(, S)] per mip level; the 3D branch repeats the full volume for every level)
        S = ns["IMAGE_SIZE"]
        images = []
        base = torch.rand(3, *([S] * D))
        for i in range(ns["MAX_MIP_LEVEL"] + 1):
            if D == 2:
                f = 2 ** i
                images.append(base.reshape(3, S // f, f, S // f, f).mean(dim=(2, 4)) if f > 1 else base)
            else:
                images.append(base)
        out[f"{tag}_image_digest"] = digest(base)
        decoder = ns["ColorDecoder"]()
        pyr = ref_fp.create_pyramid if D == 2 else ref_fp.create_pyramid_3d
        fp, _ = pyr(ns["FEATURE_PYRAMID_SIZE"], ns["FEATURE_PYRAMID_CHANNELS"], ns["FP_BITS"], "cpu", dt, ns["TF_NO_MIP"])
        for i, g_ in enumerate(fp):
            out[f"{tag}_init_grid{i}"] = g_.detach().clone()
        for k, v in decoder.state_dict().items():
            out[f"{tag}_init_sd_{k}"] = v.clone()
        optimizer = torch.optim.Adam([{"params": fp, "lr": 0.01},
                                      {"params": decoder.parameters(), "lr": 0.005}])       # image_compression.py:361-364
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=ns["NUM_EPOCHS"], eta_min=0)
        writer = _NullWriter()
        ns.update(images=images, decoder=decoder, criterion=torch.nn.MSELoss(), optimizer=optimizer,
                  scheduler=scheduler, writer=writer,
                  feature_pyramid_mip_levels_dict=ref_fp.create_pyramid_mip_levels(ns["IMAGE_SIZE"], ns["FEATURE_PYRAMID_SIZE"]))
        # record what the sampler drew, without changing it
        rec = []
        inner = ns["random_crop_dataset"]

        def spy(*a, _inner=inner, _rec=rec, **k):
            r = _inner(*a, **k)
            _rec.append((r[2], r[1].clone()))
            return r

        ns["random_crop_dataset"] = spy
        out[f"{tag}_rng_state_marker"] = torch.rand(1)          # one draw so a replay can check stream alignment
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                ns["train_models"](fp)
            finally:
                os.chdir(cwd)
        out[f"{tag}_loss"] = np.array([v for _, v in writer.scalars["Loss/train_epoch_label"]])
        out[f"{tag}_lod"] = np.array([r[0] for r in rec])
        out[f"{tag}_coord"] = np.stack([r[1].numpy() for r in rec])
        # train_models rebinds its local fp after the freeze; the caller's list keeps the frozen,
        # un-quantised tensors (image_compression.py:229-230) - that is what process_images sees.
        for i, g_ in enumerate(fp):
            out[f"{tag}_final_grid{i}"] = g_.detach().clone()
        for k, v in decoder.state_dict().items():
            out[f"{tag}_final_sd_{k}"] = v.clone()
        out[f"{tag}_cfg"] = np.array([ns["IMAGE_SIZE"], D, ns["COMPRESSION_METHOD"], ns["CROP_MIP_LEVEL"], ns["NUM_CROPS"],
                                      ns["FEATURE_PYRAMID_CHANNELS"], ns["NUM_EPOCHS"], ns["MAX_MIP_LEVEL"],
                                      int(ns["TF_NO_MIP"]), int(ns["TF_USE_TRI_PE"])])
        out[f"{tag}_uniform_rate"] = np.array(ns["UNIFORM_DISTRIBUTION_RATE"])
        # decode with the reference's decode_image on the quantised pyramid, as process_images does
        ns["PRINTLOG_PATH"] = os.devnull
        fq = ref_fp.fp_all_quantize(fp, ns["FP_BITS"])
        rec_img = ns["decode_image"](fq, decoder, 0, False)
        out[f"{tag}_decoded_mip0_digest"] = digest(rec_img)
        flat = rec_img.reshape(-1, 3)
        out[f"{tag}_decoded_mip0_rows"] = flat[:: max(1, flat.shape[0] // 512)]
        perm = (1, 2, 0) if D == 2 else (1, 2, 3, 0)
        out[f"{tag}_psnr_mip0"] = ref_utils.calculate_psnr(ref_models.quantize_to_bit(rec_img, 8),
                                                         ref_models.quantize_to_bit(images[0].permute(*perm), 8))
    save("trajectory", **out)


def g11_stored(ref_models, ref_utils, ref_fp):
    """The stored form the reference writes (image_compression.py:376-383): ``torch.save`` of the fp_savable uint8 list and of
    the decoder state_dict - committed as the two tiny .pth files themselves (data, 8 KiB) - plus the reference's own decode
    of them (decode_image on fp_load(..., float32) grids, :307-346) at 64 x 64 as the expected output."""
    ns = driver_namespace(ref_models, ref_utils, ref_fp, IMAGE_SIZE=64, CROP_MIP_LEVEL=6)
    torch.manual_seed(71)
    fp, _ = ref_fp.create_pyramid(ns["FEATURE_PYRAMID_SIZE"], 12, 8, "cpu", torch.float32, True)   # [12,17,17], [12,9,9]
    decoder = ns["ColorDecoder"]()
    ns["feature_pyramid_mip_levels_dict"] = ref_fp.create_pyramid_mip_levels(64, ns["FEATURE_PYRAMID_SIZE"])
    stored = ref_fp.fp_savable([g.detach() for g in fp], 8, torch.uint8)
    os.makedirs(OUT, exist_ok=True)
    torch.save(stored, os.path.join(OUT, "stored_feature_pyramid.pth"))
    torch.save(decoder.state_dict(), os.path.join(OUT, "stored_decoder.pth"))
    loaded = ref_fp.fp_load(stored, 8, torch.float32)
    y = ns["decode_image"](loaded, decoder, 0, pr=False)                                            # [64, 64, 3]
    save("stored_decode", y=y, y_to_bit=ref_models.quantize_to_bit(y, 8), g0_u8=stored[0], g1_u8=stored[1])


def g12_mipchain():
    """The reference's dataset chain (image_compression.py:432-440): ``transforms.Resize((S // 2^i, S // 2^i))`` + ``ToTensor`` of the PIL image.
    torchvision is not installed here; its ``Resize`` on a PIL image is ``functional.resize -> F_pil.resize -> img.resize(size[::-1], BILINEAR)``,
    so the fixture is produced by that Pillow call on a 128 x 128 crop of the reference's own sample image: the crop's codes and Pillow's levels
    64 .. 1 (each resized from the crop, like the reference resizes every level from the original)."""
    from PIL import Image
    img = Image.open("/root/reference/Projects/data/sancho_512.png").convert("RGB").crop((192, 160, 320, 288))
    arrays = {"image": np.asarray(img)}
    for i in range(1, 8):
        s = 128 // 2 ** i
        arrays[f"level_{i}"] = np.asarray(img.resize((s, s), Image.BILINEAR))
    save("mipchain", **arrays)


def main():
    torch.set_num_threads(4)
    ref_models, ref_utils, ref_fp, ref_pe = load_reference_modules()
    g1_levels(ref_fp)
    g2_pe(ref_utils, ref_pe)
    g4_g0g1_2d(ref_fp)
    g5_g0g1_3d(ref_fp)
    g6_decoder_input(ref_models, ref_utils, ref_fp)
    g7_g8_mlp_fwdbwd(ref_models, ref_utils, ref_fp)
    g8b_fwdbwd_mip0(ref_models, ref_utils, ref_fp)
    g9_codec(ref_models, ref_utils)
    g10_trajectory(ref_models, ref_utils, ref_fp)
    g11_stored(ref_models, ref_utils, ref_fp)
    g12_mipchain()
    assert not any("__pycache__" in d for d, _, _ in os.walk("/root/reference")), "bytecode written into reference"


if __name__ == "__main__":
    main()
