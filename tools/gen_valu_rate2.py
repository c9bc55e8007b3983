#!/usr/bin/env python3
"""Generates tools/valu_rate2.hip: per-opcode issue-rate micro-benchmark for gfx950.
Each kernel body = 64 instructions per loop iteration over 8 independent registers
(dependent distance 8), forced 1 block/CU through a big LDS allocation so that
waves/SIMD = blockDim/256 exactly."""
OPS = [
    # name, template using D (self reg), A, B (other vregs), S (sgpr)
    ("v_xor_b32 vvv", "v_xor_b32 {D}, {D}, {A}"),
    ("v_and_b32 vvv", "v_and_b32 {D}, {D}, {A}"),
    ("v_or_b32 vvv", "v_or_b32 {D}, {D}, {A}"),
    ("v_add_u32 vvv", "v_add_u32 {D}, {D}, {A}"),
    ("v_sub_u32 vvv", "v_sub_u32 {D}, {D}, {A}"),
    ("v_lshlrev_b32 vvv", "v_lshlrev_b32 {D}, {A}, {D}"),
    ("v_max_u32 vvv", "v_max_u32 {D}, {D}, {A}"),
    ("v_min_u32 vvv", "v_min_u32 {D}, {D}, {A}"),
    ("v_min_i32 vvv", "v_min_i32 {D}, {D}, {A}"),
    ("v_mov_b32 vv", "v_mov_b32 {D}, {A}"),
    ("v_not_b32 vv", "v_not_b32 {D}, {D}"),
    ("v_xor_b32 v,s,v", "v_xor_b32 {D}, {S}, {D}"),
    ("v_xor_b32 v,const,v", "v_xor_b32 {D}, 17, {D}"),
    ("v_xnor_b32 vvv", "v_xnor_b32 {D}, {D}, {A}"),
    ("v_bcnt_u32_b32 v,v,0", "v_bcnt_u32_b32 {D}, {D}, 0"),
    ("v_bcnt_u32_b32 v,v,v", "v_bcnt_u32_b32 {D}, {A}, {D}"),
    ("v_bcnt_u32_b32 v,s,v", "v_bcnt_u32_b32 {D}, {S}, {D}"),
    ("v_med3_u32", "v_med3_u32 {D}, {D}, {A}, {B}"),
    ("v_min3_u32", "v_min3_u32 {D}, {D}, {A}, {B}"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {D}, {D}, 1, {A}"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {D}, {D}, 1, {A}"),
    ("v_add3_u32", "v_add3_u32 {D}, {D}, {A}, {B}"),
    ("v_xad_u32", "v_xad_u32 {D}, {D}, {A}, {B}"),
    ("v_and_or_b32", "v_and_or_b32 {D}, {D}, {A}, {B}"),
    ("v_or3_b32", "v_or3_b32 {D}, {D}, {A}, {B}"),
    ("v_bfi_b32", "v_bfi_b32 {D}, {D}, {A}, {B}"),
    ("v_perm_b32", "v_perm_b32 {D}, {D}, {A}, {B}"),
    ("v_alignbit_b32", "v_alignbit_b32 {D}, {D}, {A}, {B}"),
    ("v_mad_u32_u24", "v_mad_u32_u24 {D}, {D}, {A}, {B}"),
    ("v_mul_u32_u24", "v_mul_u32_u24 {D}, {D}, {A}"),
    ("v_mul_lo_u32", "v_mul_lo_u32 {D}, {D}, {A}"),
    ("v_sad_u32", "v_sad_u32 {D}, {D}, {A}, {B}"),
    ("v_sad_u8", "v_sad_u8 {D}, {D}, {A}, {B}"),
    ("v_dot4_u32_u8", "v_dot4_u32_u8 {D}, {A}, {B}, {D}"),
    ("v_dot8_u32_u4", "v_dot8_u32_u4 {D}, {A}, {B}, {D}"),
    ("v_pk_min_u16", "v_pk_min_u16 {D}, {D}, {A}"),
    ("v_pk_add_u16", "v_pk_add_u16 {D}, {D}, {A}"),
    ("v_pk_max_u16", "v_pk_max_u16 {D}, {D}, {A}"),
    ("v_cndmask_b32", "v_cndmask_b32 {D}, {D}, {A}, vcc"),
    ("v_add_f32", "v_add_f32 {D}, {D}, {A}"),
    ("v_mul_f32", "v_mul_f32 {D}, {D}, {A}"),
    ("v_fma_f32", "v_fma_f32 {D}, {D}, {A}, {B}"),
    ("v_fmac_f32", "v_fmac_f32 {D}, {A}, {B}"),
    ("v_pk_fma_f32", "v_pk_fma_f32 {D2}, {D2}, {A2}, {B2}"),
    ("v_pk_add_f32", "v_pk_add_f32 {D2}, {D2}, {A2}"),
    ("v_max_f32", "v_max_f32 {D}, {D}, {A}"),
    # mixes (pairs alternate)
    ("mix xor_vvv+bcnt_vvv", ["v_xor_b32 {D}, {D}, {A}", "v_bcnt_u32_b32 {D}, {A}, {D}"]),
    ("mix xor_vsv+bcnt_vvv", ["v_xor_b32 {D}, {S}, {D}", "v_bcnt_u32_b32 {D}, {A}, {D}"]),
    ("mix xor_vvv+min", ["v_xor_b32 {D}, {D}, {A}", "v_min_u32 {D}, {D}, {A}"]),
    ("mix fma+bcnt", ["v_fma_f32 {D}, {D}, {A}, {B}", "v_bcnt_u32_b32 {D}, {A}, {D}"]),
    ("mix 3xor+1bcnt", ["v_xor_b32 {D}, {D}, {A}", "v_xor_b32 {D}, {D}, {B}", "v_xor_b32 {D}, {D}, {A}", "v_bcnt_u32_b32 {D}, {A}, {D}"]),
]


import os
if os.environ.get("SET", "1") == "2":
    X = "v_xor_b32 {D}, {D}, {A}"; XE = "v_xor_b32_e64 {D}, {D}, {A}"; XS = "v_xor_b32 {D}, {S}, {D}"
    BC = "v_bcnt_u32_b32 {D}, {A}, {D}"; FMA = "v_fma_f32 {D}, {D}, {A}, {B}"; MIN = "v_min_u32 {D}, {D}, {A}"
    ADDF = "v_add_f32 {D}, {D}, {A}"; MULF = "v_mul_f32 {D}, {D}, {A}"; FMAC = "v_fmac_f32 {D}, {A}, {B}"
    MOV = "v_mov_b32 {D}, {A}"; AND = "v_and_b32 {D}, {D}, {A}"; ADDU = "v_add_u32 {D}, {D}, {A}"
    LOR = "v_lshl_or_b32 {D}, {D}, 1, {A}"; MED = "v_med3_u32 {D}, {D}, {A}, {B}"; XC = "v_xor_b32 {D}, 17, {D}"
    ADDFE = "v_add_f32_e64 {D}, {D}, {A}"; MINE = "v_min_u32_e64 {D}, {D}, {A}"
    OPS = [
        ("xor_e64 alone", XE), ("add_f32_e64 alone", ADDFE), ("min_e64 alone", MINE),
        ("fma+xor", [FMA, X]), ("fma+min", [FMA, MIN]), ("fma+lshl_or", [FMA, LOR]), ("fma+med3", [FMA, MED]),
        ("addf+bcnt", [ADDF, BC]), ("mulf+bcnt", [MULF, BC]), ("fmac+bcnt", [FMAC, BC]), ("mov+bcnt", [MOV, BC]),
        ("and+bcnt", [AND, BC]), ("addu+bcnt", [ADDU, BC]), ("xorconst+bcnt", [XC, BC]), ("xor_e64+bcnt", [XE, BC]),
        ("addf+xor", [ADDF, X]), ("addf+min", [ADDF, MIN]), ("addf_e64+bcnt", [ADDFE, BC]),
        ("2fma+1bcnt", [FMA, FMA, BC]), ("1fma+2bcnt", [FMA, BC, BC]), ("1fma+3bcnt", [FMA, BC, BC, BC]),
        ("fma+xor+bcnt", [FMA, X, BC]), ("fma+xor+bcnt+xor", [FMA, X, BC, X]),
        ("8xor then 8bcnt", [X]*8 + [BC]*8), ("8xor_e64 then 8bcnt", [XE]*8 + [BC]*8),
        ("xor_s+bcnt", [XS, BC]),
        ("4x(xor,bcnt),4fma", [X, BC, X, BC, X, BC, X, BC, FMA, FMA, FMA, FMA]),
    ]

if os.environ.get("SET", "1") == "3":
    X = "v_xor_b32 {D}, {D}, {A}"; XS = "v_xor_b32 {D}, {S}, {D}"
    BC = "v_bcnt_u32_b32 {D}, {A}, {D}"; FMA = "v_fma_f32 {D}, {D}, {A}, {B}"; MIN = "v_min_u32 {D}, {D}, {A}"
    ADDF = "v_add_f32 {D}, {D}, {A}"; MOV = "v_mov_b32 {D}, {A}"; NOT = "v_not_b32 {D}, {D}"
    LOR = "v_lshl_or_b32 {D}, {D}, 1, {A}"; MED = "v_med3_u32 {D}, {D}, {A}, {B}"
    NOP = "s_nop 0"
    OPS = [
        ("[X,C]", [X, BC]), ("[C,X]", [BC, X]), ("[X,X,C,C]", [X, X, BC, BC]),
        ("[X,C]x2+F", [X, BC] * 2 + [FMA]), ("[X,C]x4+F", [X, BC] * 4 + [FMA]), ("[X,C]x8+F", [X, BC] * 8 + [FMA]),
        ("[X,C]x4+MOV", [X, BC] * 4 + [MOV]), ("[X,C]x8+MOV", [X, BC] * 8 + [MOV]), ("[X,C]x4+ADDF", [X, BC] * 4 + [ADDF]),
        ("[X,C]x2+MOV", [X, BC] * 2 + [MOV]), ("[X,C,MOV]", [X, BC, MOV]), ("[X,MOV,C,MOV]", [X, MOV, BC, MOV]),
        ("[XS,C,F]", [XS, BC, FMA]), ("[XS,C]x4+F", [XS, BC] * 4 + [FMA]), ("[XS,C,MOV]", [XS, BC, MOV]),
        ("[C]x4+F", [BC] * 4 + [FMA]), ("[C]x2+MOV", [BC] * 2 + [MOV]), ("[C,MOV]", [BC, MOV]), ("[MIN,MOV]", [MIN, MOV]),
        ("[X]x7+C", [X] * 7 + [BC]), ("[X]x3+MOV", [X] * 3 + [MOV]),
        ("[X,C]x4+NOP", [X, BC] * 4 + [NOP]), ("[X,C,NOP]", [X, BC, NOP]),
        ("real19 vgpr", [X, BC] * 8 + [LOR, MED, MIN]), ("real19 sgpr", [XS, BC] * 8 + [LOR, MED, MIN]),
        ("real19 vgpr+mov/2", [X, BC, MOV] * 8 + [LOR, MOV, MED, MOV, MIN, MOV]),
        ("real19 sgpr+mov/2", [XS, BC, MOV] * 8 + [LOR, MOV, MED, MOV, MIN, MOV]),
        ("real19 vgpr+mov/4", [X, BC, X, BC, MOV] * 4 + [LOR, MED, MOV, MIN]),
    ]

def body(tmpl):
    tl = tmpl if isinstance(tmpl, list) else [tmpl]
    lines = []
    n = 0
    for rep in range(8):
        for r in range(8):
            t = tl[n % len(tl)]
            n += 1
            if "{D2}" in t:
                # register pairs: use 4 pairs a[0:1].. via 64-bit operands %0..%3, others %4..%7
                d = "%%%d" % (r % 4); a = "%%%d" % (4 + (r + rep) % 4); b = "%%%d" % (4 + (r + rep + 1) % 4)
                lines.append(t.format(D2=d, A2=a, B2=b))
            else:
                d = "%%%d" % r; a = "%%%d" % (8 + (r + rep) % 8); b = "%%%d" % (8 + (r + rep + 3) % 8)
                s = "%%%d" % (16 + (r + rep) % 4)
                lines.append(t.format(D=d, A=a, B=b, S=s))
    return lines

out = []
out.append("// GENERATED by tools/gen_valu_rate2.py -- do not edit.\n")
out.append(open(__file__.replace("gen_valu_rate2.py", "valu_rate2_head.inc")).read())
for i, (name, tmpl) in enumerate(OPS):
    lines = body(tmpl)
    pk = any("{D2}" in t for t in (tmpl if isinstance(tmpl, list) else [tmpl]))
    out.append("template <> __device__ __forceinline__ void body<%d>(REGS) {\n" % i)
    asm = "\n".join('      "%s\\n"' % l for l in lines)
    if pk:
        out.append("  asm volatile(\n%s\n      : \"+v\"(p0), \"+v\"(p1), \"+v\"(p2), \"+v\"(p3)\n      : \"v\"(q0), \"v\"(q1), \"v\"(q2), \"v\"(q3));\n}\n" % asm)
    else:
        out.append("  asm volatile(\n%s\n      : \"+v\"(a0), \"+v\"(a1), \"+v\"(a2), \"+v\"(a3), \"+v\"(a4), \"+v\"(a5), \"+v\"(a6), \"+v\"(a7)\n"
                   "      : \"v\"(b0), \"v\"(b1), \"v\"(b2), \"v\"(b3), \"v\"(b4), \"v\"(b5), \"v\"(b6), \"v\"(b7), \"s\"(s0), \"s\"(s1), \"s\"(s2), \"s\"(s3) : \"vcc\");\n}\n" % asm)
out.append(open(__file__.replace("gen_valu_rate2.py", "valu_rate2_tail.inc")).read())
out.append("static const char* op_names[] = {%s};\n" % ", ".join('"%s"' % n for n, _ in OPS))
out.append("#define N_OPS %d\n" % len(OPS))
out.append("static kern_t kernels[N_OPS] = {%s};\n" % ", ".join("rate_kernel<%d>" % i for i in range(len(OPS))))
out.append(open(__file__.replace("gen_valu_rate2.py", "valu_rate2_main.inc")).read())
open(__file__.replace("gen_valu_rate2.py", "valu_rate2.hip"), "w").write("".join(out))
