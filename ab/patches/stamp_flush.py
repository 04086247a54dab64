"""diagnostic patch (not product): sub-stamps of the flush in fused_q16.hpp - phases 5 (until the G1 lane combine is through), 6 (group sum: LDS +
two barriers), 7 (pre-add), 13 (atomics) - in place of the prologue's stamps.  python ab/patches/stamp_flush.py apply | revert; build with
ab/q16/mk.sh stf -DNIC_STAMPS -DNIC_STAMP_FLUSH and read with ab/q16/stamps_sweep.py."""
import subprocess, sys
p = "neural_image_compression_v2_amd/csrc/fused_q16.hpp"
if sys.argv[1] == "revert":
    raise SystemExit("revert by hand (git diff): a checkout would drop other uncommitted edits of the file")
    sys.exit(0)
s = open(p).read()
def rep(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, 1)
i = s.index("    if (NL == 3) { stamp_sum[9] = stamp_t1 - stamp_t0;")
j = s.index("\n", i)
s = s[:i] + "#ifndef NIC_STAMP_FLUSH\n" + s[i:j + 1] + "#endif\n" + s[j + 1:]
rep("""            combine_g1_lanes_q<Q>(g1s, blk_off1, blk, ln, lw, packed, pk, pk_lc);
            bool flush = true;""", """            combine_g1_lanes_q<Q>(g1s, blk_off1, blk, ln, lw, packed, pk, pk_lc);
#ifdef NIC_STAMP_FLUSH
            STAMP(5);
#endif
            bool flush = true;""")
rep("""                flush = leader == wave;
            }
            constexpr bool SHARE_XY = !Q::TETRA;""", """                flush = leader == wave;
            }
#ifdef NIC_STAMP_FLUSH
            STAMP(6);
#endif
            constexpr bool SHARE_XY = !Q::TETRA;""")
rep("""            if (flush) {
                const uint32_t pb0 = (uint32_t)p.g0.plane * 4u, pb1 = (uint32_t)p.g1.plane * 4u;""", """#ifdef NIC_STAMP_FLUSH
            STAMP(7);
#endif
            if (flush) {
                const uint32_t pb0 = (uint32_t)p.g0.plane * 4u, pb1 = (uint32_t)p.g1.plane * 4u;""")
open(p, "w").write(s)
