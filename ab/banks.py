"""diagnostic: LDS bank-conflict factors of the access patterns of fused_train16_kernel / fused_q16_kernel under the bank rules of
/opt/skills/guides/MI355X_MICROARCH.md (LDS): per instruction, fixed lane groups; lanes of one group conflict when they touch the same bank with
different addresses.  Prints for every pattern the LDS-array cycles (conflict-free minimum in brackets).   python ab/banks.py"""
import itertools

G_B128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          [32 + x for x in list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))], [32 + x for x in list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]]
G_HALF = [list(range(0, 32)), list(range(32, 64))]
G_W128 = [list(range(8 * k, 8 * k + 8)) for k in range(8)]
G_W64 = [list(range(16 * k, 16 * k + 16)) for k in range(4)]


def cycles(addr, nbytes, groups, nbanks):
    """addr(lane) -> byte address; every lane touches nbytes / 4 consecutive dwords"""
    tot = 0
    for grp in groups:
        per_bank = {}
        for l in grp:
            a = addr(l)
            for k in range(nbytes // 4):
                dw = a // 4 + k
                per_bank.setdefault(dw % nbanks, set()).add(dw)
        tot += max(len(v) for v in per_bank.values())
    return tot, len(groups)


def lane(l):
    return dict(n16=l & 15, g=l >> 4, q4=(l & 15) >> 2, p4=l & 3, h32=l >> 5, cg=(l >> 4) & 1)


def report(name, addr, nbytes, groups, nbanks):
    c, m = cycles(addr, nbytes, groups, nbanks)
    print(f"{name:78s} {c:3d} cycles ({m} conflict-free){'   <-- ' + str(round(c / m, 2)) + ' x' if c > m else ''}")


for fam, LD1, LDH, LDZ, LDX in (("train16", 80, 80, 72, 88), ("q16 2D (NL 3 / 5)", 80, 80, 72, 88), ("q16 method 3", 144, 80, 72, 136)):
    print(f"== {fam}: LD1 {LD1} LDH {LDH} LDZ {LDZ} LDX {LDX} (bf16 elements)")
    E = 2
    report("forward A fragments, 64-column images: ds_read_b128 row n16 + 16 t, columns 8 g", lambda l: (lane(l)["n16"] * LDH + 8 * lane(l)["g"]) * E, 16, G_B128, 64)
    report("forward A fragments of W1: ds_read_b128", lambda l: (lane(l)["n16"] * LD1 + 8 * lane(l)["g"]) * E, 16, G_B128, 64)
    report("fragment stores, A / DZ images: ds_write_b128 row n16, columns 8 g", lambda l: (lane(l)["n16"] * LDZ + 8 * lane(l)["g"]) * E, 16, G_W128, 32)
    report("fragment stores, X image: ds_write_b128", lambda l: (lane(l)["n16"] * LDX + 8 * lane(l)["g"]) * E, 16, G_W128, 32)
    report("transposed weight reads (dA = W^T dZ): ds_read_b64_tr_b16 row 4 g + q4, columns 8 p4", lambda l: ((4 * lane(l)["g"] + lane(l)["q4"]) * LDH + 8 * lane(l)["p4"]) * E, 8, G_HALF, 64)
    report("  .. of W1 (dX)", lambda l: ((4 * lane(l)["g"] + lane(l)["q4"]) * LD1 + 8 * lane(l)["p4"]) * E, 8, G_HALF, 64)
    report("4x4x4 operands of the DZ image: ds_read_b64 row 4 q4, columns 16 g + 4 p4", lambda l: (4 * lane(l)["q4"] * LDZ + 16 * lane(l)["g"] + 4 * lane(l)["p4"]) * E, 8, G_HALF, 64)
    report("32x32x16 operands, DZ / A images: ds_read_b64_tr_b16 row 4 q4 + 2 h32, columns 16 cg + 4 p4", lambda l: ((4 * lane(l)["q4"] + 2 * lane(l)["h32"]) * LDZ + 16 * lane(l)["cg"] + 4 * lane(l)["p4"]) * E, 8, G_HALF, 64)
    report("  .. X image", lambda l: ((4 * lane(l)["q4"] + 2 * lane(l)["h32"]) * LDX + 16 * lane(l)["cg"] + 4 * lane(l)["p4"]) * E, 8, G_HALF, 64)
    report("16x16 tail operands: row 4 q4 + 2 (g & 1), columns 4 p4, source wave by g >> 1 (SPW apart)", lambda l: ((lane(l)["g"] >> 1) * 5000 * 0 + (4 * lane(l)["q4"] + 2 * (lane(l)["g"] & 1)) * LDZ + 4 * lane(l)["p4"]) * E, 8, G_HALF, 64)
    report("dY / dZ_out image: ds_write_b16 [c][16] rows 32 B apart, lanes n16 (3 rows)", lambda l: (lane(l)["n16"]) * E, 4, G_HALF, 32)
