"""Generates shift_probe_*.s: hand-written gfx950 kernels, each allocating exactly N VGPRs, that execute one reference
instruction with all operands in the middle of the allocation and one test instruction with an operand in the LAST allocated
VGPR(s), every round, and count the rounds in which the two results differ (plus the round of the first difference and the
wrong value).  Usage: python make_shift_probe.py"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
HEAD = open(os.path.join(HERE, "make_alias_probe.py")).read()
TEMPLATE = HEAD[HEAD.index('TEMPLATE = """') + len('TEMPLATE = """'):HEAD.index('"""\n\n\ndef gen')]

# name -> (N, setup lines, reference instruction writing v[6:7], test instruction writing v[8:9] (or custom), between)
def tests():
    t = {}
    for n in (24, 40, 64):
        top = n - 1
        t["lshr_count_top_%d" % n] = (n, ["v_mov_b32_e32 v%d, 3" % top], "v_lshrrev_b64 v[6:7], v10, v[4:5]", "v_lshrrev_b64 v[8:9], v%d, v[4:5]" % top, [])
    n, top = 24, 23
    S = lambda name, setup, ref, test, between=(): t.__setitem__(name, (n, list(setup), ref, test, list(between)))
    S("lshr_count_top_minus1", ["v_mov_b32_e32 v22, 3"], "v_lshrrev_b64 v[6:7], v10, v[4:5]", "v_lshrrev_b64 v[8:9], v22, v[4:5]")
    S("lshr_count_top_valu_between", ["v_mov_b32_e32 v23, 3"], "v_lshrrev_b64 v[6:7], v10, v[4:5]", "v_lshrrev_b64 v[8:9], v23, v[4:5]", ["v_mov_b32_e32 v18, v1", "v_mov_b32_e32 v19, v1"])
    S("lshr_count_top_test_first", ["v_mov_b32_e32 v23, 3"], "v_lshrrev_b64 v[6:7], v10, v[4:5]", "v_lshrrev_b64 v[8:9], v23, v[4:5]", ["SWAP"])
    S("lshr_count_top_alone", ["v_mov_b32_e32 v23, 3"], "v_lshrrev_b32_e32 v6, v10, v4\n\tv_mov_b32_e32 v7, 0", "v_lshrrev_b64 v[8:9], v23, v[4:5]\n\tv_mov_b32_e32 v9, 0\n\tv_and_b32_e32 v6, 0x1fffffff, v6\n\tv_and_b32_e32 v8, 0x1fffffff, v8")
    S("lshr_src_pair_top", ["v_mov_b32_e32 v22, v4", "v_mov_b32_e32 v23, v5"], "v_lshrrev_b64 v[6:7], v10, v[4:5]", "v_lshrrev_b64 v[8:9], v10, v[22:23]")
    S("lshr_dst_pair_top", [], "v_lshrrev_b64 v[6:7], v10, v[4:5]", "v_lshrrev_b64 v[22:23], v10, v[4:5]\n\tv_mov_b32_e32 v8, v22\n\tv_mov_b32_e32 v9, v23")
    S("lshr32_count_top", ["v_mov_b32_e32 v23, 3"], "v_lshrrev_b32_e32 v6, v10, v4\n\tv_mov_b32_e32 v7, 0", "v_lshrrev_b32_e32 v8, v23, v4\n\tv_mov_b32_e32 v9, 0")
    S("lshl_add_u64_src2_top", ["v_mov_b32_e32 v22, v4", "v_mov_b32_e32 v23, v5", "v_mov_b32_e32 v20, v4", "v_mov_b32_e32 v21, v5"],
      "v_lshl_add_u64 v[6:7], v[4:5], 0, v[20:21]", "v_lshl_add_u64 v[8:9], v[4:5], 0, v[22:23]")
    S("mad_u64_u32_src1_top", ["v_mov_b32_e32 v23, 3"], "v_mad_u64_u32 v[6:7], s[8:9], v4, v10, v[4:5]", "v_mad_u64_u32 v[8:9], s[8:9], v4, v23, v[4:5]")
    S("add_f64_src_top", ["v_mov_b32_e32 v22, v4", "v_mov_b32_e32 v23, 0x3ff00000", "v_mov_b32_e32 v20, v4", "v_mov_b32_e32 v21, 0x3ff00000"],
      "v_add_f64 v[6:7], v[20:21], v[20:21]", "v_add_f64 v[8:9], v[22:23], v[22:23]")
    S("mul_lo_u32_src_top", ["v_mov_b32_e32 v23, 3"], "v_mul_lo_u32 v6, v4, v10\n\tv_mov_b32_e32 v7, 0", "v_mul_lo_u32 v8, v4, v23\n\tv_mov_b32_e32 v9, 0")
    S("alignbit_count_top", ["v_mov_b32_e32 v23, 3"], "v_alignbit_b32 v6, v5, v4, v10\n\tv_mov_b32_e32 v7, 0", "v_alignbit_b32 v8, v5, v4, v23\n\tv_mov_b32_e32 v9, 0")
    # the last USED register below an allocation boundary (22 or 23 registers used, 24 allocated): is the limit the allocation's?
    for nn in (22, 23):
        t["lshr_count_lastused_%d" % nn] = (nn, ["v_mov_b32_e32 v%d, 3" % (nn - 1)], "v_lshrrev_b64 v[6:7], v10, v[4:5]", "v_lshrrev_b64 v[8:9], v%d, v[4:5]" % (nn - 1), [])
    S("lshl_count_top", ["v_mov_b32_e32 v23, 3"], "v_lshlrev_b64 v[6:7], v10, v[4:5]", "v_lshlrev_b64 v[8:9], v23, v[4:5]")
    S("ashr_count_top", ["v_mov_b32_e32 v23, 3"], "v_ashrrev_i64 v[6:7], v10, v[4:5]", "v_ashrrev_i64 v[8:9], v23, v[4:5]")
    S("cvt_f64_u32_src_top", ["v_mov_b32_e32 v23, v4", "v_mov_b32_e32 v20, v4"], "v_cvt_f64_u32_e32 v[6:7], v20", "v_cvt_f64_u32_e32 v[8:9], v23")
    S("cvt_f64_i32_src_top", ["v_mov_b32_e32 v23, v4", "v_mov_b32_e32 v20, v4"], "v_cvt_f64_i32_e32 v[6:7], v20", "v_cvt_f64_i32_e32 v[8:9], v23")
    S("cvt_f64_f32_src_top", ["v_cvt_f32_u32_e32 v23, v1", "v_cvt_f32_u32_e32 v20, v1"], "v_cvt_f64_f32_e32 v[6:7], v20", "v_cvt_f64_f32_e32 v[8:9], v23")
    S("ldexp_f64_exp_top", ["v_mov_b32_e32 v23, 3", "v_mov_b32_e32 v20, v4", "v_mov_b32_e32 v21, 0x3ff00000"], "v_ldexp_f64 v[6:7], v[20:21], v10", "v_ldexp_f64 v[8:9], v[20:21], v23")
    S("mad_u64_u32_src0_top", ["v_mov_b32_e32 v23, 3"], "v_mad_u64_u32 v[6:7], s[8:9], v10, v4, v[4:5]", "v_mad_u64_u32 v[8:9], s[8:9], v23, v4, v[4:5]")
    S("mad_i64_i32_src0_top", ["v_mov_b32_e32 v23, 3"], "v_mad_i64_i32 v[6:7], s[8:9], v10, v4, v[4:5]", "v_mad_i64_i32 v[8:9], s[8:9], v23, v4, v[4:5]")
    S("cvt_f32_f64_dst_top", ["v_mov_b32_e32 v20, v4", "v_mov_b32_e32 v21, 0x3ff00000"], "v_cvt_f32_f64_e32 v6, v[20:21]\n\tv_mov_b32_e32 v7, 0", "v_cvt_f32_f64_e32 v23, v[20:21]\n\tv_mov_b32_e32 v8, v23\n\tv_mov_b32_e32 v9, 0")
    S("cvt_u32_f64_dst_top", ["v_mov_b32_e32 v20, v4", "v_mov_b32_e32 v21, 0x40f00000"], "v_cvt_u32_f64_e32 v6, v[20:21]\n\tv_mov_b32_e32 v7, 0", "v_cvt_u32_f64_e32 v23, v[20:21]\n\tv_mov_b32_e32 v8, v23\n\tv_mov_b32_e32 v9, 0")
    S("frexp_exp_i32_f64_dst_top", ["v_mov_b32_e32 v20, v4", "v_mov_b32_e32 v21, 0x40f00000"], "v_frexp_exp_i32_f64_e32 v6, v[20:21]\n\tv_mov_b32_e32 v7, 0", "v_frexp_exp_i32_f64_e32 v23, v[20:21]\n\tv_mov_b32_e32 v8, v23\n\tv_mov_b32_e32 v9, 0")
    return t


def gen(n, setup, ref, test, between):
    body = TEMPLATE
    a = body.index("\tv_lshl_add_u32 v1, s2, 8, v0")
    b = body.index("\ts_endpgm")
    first, second = (test, ref) if between == ["SWAP"] else (ref, test)
    mid = [] if between == ["SWAP"] else between
    code = "\n".join(["\tv_lshl_add_u32 v1, s2, 8, v0                ; tag = workgroup * 256 + thread",
                      "\tv_or_b32_e32 v4, 0x80000000, v1             ; value.lo",
                      "\tv_mov_b32_e32 v5, 0x1234                    ; value.hi",
                      "\tv_mov_b32_e32 v10, 3                        ; the count / small operand, middle of the allocation",
                      "\tv_mov_b32_e32 v14, 0                        ; rounds in which the results differed",
                      "\tv_mov_b32_e32 v15, 0                        ; round of the first difference + 1",
                      "\tv_mov_b32_e32 v16, 0                        ; last wrong result, low word",
                      "\tv_mov_b32_e32 v17, 0                        ; last wrong result, high word"]
                     + ["\t" + s for s in setup]
                     + ["\ts_mov_b32 s7, 0", "\ts_waitcnt lgkmcnt(0)", ".Lround:", "\t" + first] + ["\t" + m for m in mid] + ["\t" + second,
                        "\tv_cmp_ne_u64_e32 vcc, v[6:7], v[8:9]", "\ts_nop 1",
                        "\tv_cndmask_b32_e32 v16, v16, v8, vcc", "\tv_cndmask_b32_e32 v17, v17, v9, vcc",
                        "\tv_cmp_eq_u32_e64 s[10:11], 0, v15", "\ts_and_b64 s[10:11], s[10:11], vcc", "\ts_add_u32 s12, s7, 1", "\tv_mov_b32_e32 v18, s12",
                        "\tv_cndmask_b32_e64 v15, v15, v18, s[10:11]",
                        "\tv_addc_co_u32_e32 v14, vcc, 0, v14, vcc",
                        "\ts_sleep 2", "\ts_add_u32 s7, s7, 1", "\ts_cmp_lt_u32 s7, s6", "\ts_cbranch_scc1 .Lround",
                        "\t; record: tag, rounds wrong, first wrong round + 1, 0, wrong.lo, wrong.hi, value.lo, 0",
                        "\tv_mov_b32_e32 v2, v1", "\tv_mov_b32_e32 v3, v14", "\tv_mov_b32_e32 v6, v16", "\tv_mov_b32_e32 v7, v17", "\tv_mov_b32_e32 v8, v4",
                        "\tv_mov_b32_e32 v9, 0", "\tv_mov_b32_e32 v4, v15", "\tv_mov_b32_e32 v5, 0", "\tv_lshlrev_b32_e32 v10, 5, v1", "\tv_mov_b32_e32 v11, 0",
                        "\tv_lshl_add_u64 v[12:13], v[10:11], 0, s[4:5]", "\tglobal_store_dwordx4 v[12:13], v[2:5], off",
                        "\tglobal_store_dwordx4 v[12:13], v[6:9], off offset:16", ""])
    body = body[:a] + code + body[b:]
    body = body.replace("alias_probe", "shift_probe").replace("VGPR aliasing probe", "operand-in-the-last-VGPR probe")
    return body.format(n=n, acc=(n + 3) // 4 * 4, init="", check="", top=n - 1, top1=n - 2, topx="0")


if __name__ == "__main__":
    for f in os.listdir(HERE):
        if f.startswith("shift_probe_") and f.endswith(".s"):
            os.remove(os.path.join(HERE, f))
    for name, (n, setup, ref, test, between) in tests().items():
        open(os.path.join(HERE, "shift_probe_%s.s" % name), "w").write(gen(n, setup, ref, test, between))
