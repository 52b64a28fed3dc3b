#!/usr/bin/env python3
"""oracle/build_ref.py -- build oracle/_ref/libovrfsr_ref.so from the reference sources WHERE THEY LIE.

TEST INFRASTRUCTURE ONLY.  The reference (/root/reference, read-only) ships its filters as HLSL
compute shaders, which no tool in this image can compile.  This recipe turns the reference's own
text into C++ mechanically and compiles it against oracle/hlsl_shim.hpp:

  * fsr/ffx_a.h          -> the ~20 one-line helpers the F bodies call (HLSL/GPU sections only)
  * fsr/ffx_fsr1.h       -> the `#if defined(A_GPU)&&defined(FSR_EASU_F)` and `...FSR_RCAS_F` blocks
  * fsr/fsr_easu.hlsl, fsr/fsr_rcas.hlsl -> cbuffer / resources / callbacks / main, verbatim
  * nis/NIS_Scaler.h, nis/NIS_Upscale.hlsl, nis/NIS_Sharpen.hlsl -> NVScaler / NVSharpen + main

The only edits are syntactic (HLSL -> C++): `inout T x`/`out T x` -> `T& x`, register/semantic
annotations dropped, `cbuffer {}` -> namespace-scope variables, `main` -> `cs_main`,
`groupshared` -> static, `[unroll]` dropped.  No arithmetic is touched.  The generated C++ is an
intermediate under oracle/_ref/ (git-ignored) and is deleted after compilation unless --keep.

The constant-setup code (FsrEasuCon, FsrRcasCon, NVScalerUpdateConfig, coefficient tables) is real
C/C++ in the reference and is compiled untouched by oracle/ref_consts.cpp with `#define A_CPU`.

Usage: python oracle/build_ref.py [--ref /root/reference] [--keep]
Exit status 0 and no output file if the reference tree is absent (GPU box): the prebuilt .so that
travelled with the snapshot is used instead.
"""
import argparse
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

TYPE_MACROS = r"""
// type layer of fsr/ffx_a.h:1050-1084 (non-6.2 HLSL flavour), restated for C++
#define AP1 bool
#define AF1 float
#define AF2 float2
#define AF3 float3
#define AF4 float4
#define AU1 uint
#define AU2 uint2
#define AU3 uint3
#define AU4 uint4
#define ASU1 int
#define ASU2 int2
#define AF1_AU1(x) asfloat(AU1(x))
#define AF2_AU2(x) asfloat(AU2(x))
#define AU1_AF1(x) asuint(AF1(x))
#define AU2_AF2(x) asuint(AF2(x))
"""

FFX_A_HELPERS = [
    "AF1_x", "AF2_x", "AF3_x", "AF4_x", "AU1_x", "AU2_x",
    "ABfe", "ABfiM", "AMax3F1", "AMax3F3", "AMin3F1", "AMin3F3", "ARcpF1", "ASatF1",
    "APrxLoRcpF1", "APrxMedRcpF1", "APrxLoRsqF1", "ARmp8x8",
]
FFX_A_MACROS = ["AF1_", "AF2_", "AF3_", "AF4_", "AU1_", "AU2_"]


def read(path):
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        return f.read().replace("\r\n", "\n")


def extract_ffx_a(text):
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if "#if defined(A_HLSL) && defined(A_GPU)" in l)
    out = []
    body = lines[start:]
    for name in FFX_A_MACROS:
        pat = re.compile(r"^\s*#define\s+%s\(a\)\s" % re.escape(name))
        hit = [l for l in body if pat.match(l)]
        if not hit:
            raise SystemExit("build_ref: macro %s not found in ffx_a.h" % name)
        out.append(hit[0].strip())
    for name in FFX_A_HELPERS:
        pat = re.compile(r"^\s*A[A-Z]+[1-4]\s+%s\(" % re.escape(name))
        hit = [l for l in body if pat.match(l)]
        if not hit:
            raise SystemExit("build_ref: helper %s not found in ffx_a.h" % name)
        out.append("inline " + hit[0].strip())
    return "\n".join(out)


def extract_block(text, opener):
    """Text between the `#if ...opener...` line and its matching #endif (exclusive)."""
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.strip().startswith("#if") and opener in l)
    depth, i = 1, start + 1
    while depth:
        s = lines[i].strip()
        if s.startswith("#if"):
            depth += 1
        elif s.startswith("#endif"):
            depth -= 1
        i += 1
    return "\n".join(lines[start + 1:i - 1])


def hlsl_params_to_cpp(text):
    text = re.sub(r"\binout\s+(\w+)\s+(\w+)", r"\1& \2", text)
    text = re.sub(r"\bout\s+(A[A-Z]+[1-4])\s+(\w+)", r"\1& \2", text)
    return text


def hlsl_entry_to_cpp(text, includes):
    """fsr_*.hlsl / NIS_*.hlsl -> C++ (syntax only)."""
    def inc(m):
        name = m.group(1)
        if name not in includes:
            raise SystemExit("build_ref: unexpected include %s" % name)
        return includes[name]
    text = re.sub(r'#include\s+"([^"]+)"', inc, text)
    text = re.sub(r"cbuffer\s+\w+\s*:\s*register\(\w+\)\s*\{", "namespace {", text)
    text = re.sub(r":\s*register\(\w+\)", "", text)
    text = re.sub(r"Texture2D<\w+>", "Texture2D", text)
    text = re.sub(r"RWTexture2D<[\w\s]+>", "RWTexture2D", text)
    text = re.sub(r"\[numthreads\([^\]]*\)\]", "", text)
    text = re.sub(r":\s*SV_\w+", "", text)
    text = re.sub(r"\bvoid\s+main\s*\(", "void cs_main(", text)
    return hlsl_params_to_cpp(text)


def gen_fsr(ref):
    fsr = os.path.join(ref, "src", "fsr")
    ffx_a = read(os.path.join(fsr, "ffx_a.h"))
    ffx_fsr1 = read(os.path.join(fsr, "ffx_fsr1.h"))
    helpers = extract_ffx_a(ffx_a)
    easu_block = hlsl_params_to_cpp(extract_block(ffx_fsr1, "defined(A_GPU)&&defined(FSR_EASU_F)"))
    rcas_limit = next(l for l in ffx_fsr1.split("\n") if l.startswith("#define FSR_RCAS_LIMIT"))
    rcas_block = hlsl_params_to_cpp(extract_block(ffx_fsr1, "defined(A_GPU)&&defined(FSR_RCAS_F)"))
    easu_main = hlsl_entry_to_cpp(read(os.path.join(fsr, "fsr_easu.hlsl")),
                                  {"ffx_a.h": "", "ffx_fsr1.h": easu_block})
    rcas_main = hlsl_entry_to_cpp(read(os.path.join(fsr, "fsr_rcas.hlsl")),
                                  {"ffx_a.h": "", "ffx_fsr1.h": rcas_limit + "\n" + rcas_block})
    return "\n".join([
        "// GENERATED by oracle/build_ref.py from the reference's fsr/ sources. Build intermediate; never commit.",
        '#include "../hlsl_shim.hpp"',
        TYPE_MACROS,
        "namespace hlsl { namespace ref {",
        helpers,
        "namespace easu {", easu_main, "}",
        "namespace rcas {", rcas_main, "}",
        "}}",
        '#include "../ref_fsr_driver.inc"',
        "",
    ])


def gen_nis(ref):
    nis = os.path.join(ref, "src", "nis")
    scaler = read(os.path.join(nis, "NIS_Scaler.h"))

    def nis_to_cpp(text):
        # HLSL floating literals without a suffix are `float`; in C++ they would be `double` and drag whole
        # expressions into double arithmetic.  Suffix them (comments/preprocessor integers are unaffected).
        text = re.sub(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?)(?![\w.])", r"\1f", text)
        text = re.sub(r"\bgroupshared\b", "static", text)
        text = re.sub(r"#define\s+NIS_UNROLL\s+\[unroll\]", "#define NIS_UNROLL", text)
        # C++ needs the array-typed prototypes as they are; float4 casts `(NVF4)x` are fine.
        return text

    def one(fname, ns):
        src = read(os.path.join(nis, fname))
        # each entry file #defines its own NIS_SCALER / block sizes before including the header
        body = hlsl_entry_to_cpp(nis_to_cpp(src), {"NIS_Scaler.h": nis_to_cpp(scaler)})
        body = re.sub(r"\bTexture2D\s+(\w+)\s*;", r"Texture2D \1;", body)
        return "namespace %s {\n%s\n}\n" % (ns, body)

    undef = "\n".join("#undef " + m for m in [
        "NIS_SCALER", "NIS_HDR_MODE", "NIS_BLOCK_WIDTH", "NIS_BLOCK_HEIGHT", "NIS_THREAD_GROUP_SIZE",
        "NIS_HDR_MODE_NONE", "NIS_HDR_MODE_LINEAR", "NIS_HDR_MODE_PQ", "kHDRCompressionFactor",
        "NIS_VIEWPORT_SUPPORT", "NIS_USE_HALF_PRECISION", "NIS_HLSL_6_2", "NIS_SCALE_INT",
        "NIS_SCALE_FLOAT", "NIS_UNROLL", "NIS_TEXTURE_GATHER", "kPhaseCount", "kFilterSize",
        "kSupportSize", "kPadSize", "kTileSize", "blockDim", "kNumPixelsX", "kNumPixelsY"])
    return "\n".join([
        "// GENERATED by oracle/build_ref.py from the reference's nis/ sources. Build intermediate; never commit.",
        '#include "../hlsl_shim.hpp"',
        "namespace hlsl { namespace ref {",
        one("NIS_Upscale.hlsl", "nis_upscale"),
        undef,
        one("NIS_Sharpen.hlsl", "nis_sharpen"),
        "}}",
        '#include "../ref_nis_driver.inc"',
        "",
    ])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--keep", action="store_true", help="keep the generated C++ intermediates")
    ap.add_argument("--no-nis", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(args.ref, "src", "fsr")):
        print("build_ref: %s not present; keeping any prebuilt oracle/_ref" % args.ref)
        return 0
    os.makedirs(OUT, exist_ok=True)
    gens = {"gen_fsr_hlsl.cpp": gen_fsr(args.ref)}
    if not args.no_nis and os.path.exists(os.path.join(HERE, "ref_nis_driver.inc")):
        gens["gen_nis_hlsl.cpp"] = gen_nis(args.ref)
    srcs = []
    for name, text in gens.items():
        p = os.path.join(OUT, name)
        with open(p, "w") as f:
            f.write(text)
        srcs.append(p)
    srcs.append(os.path.join(HERE, "ref_consts.cpp"))
    so = os.path.join(OUT, "libovrfsr_ref.so")
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-fvisibility=hidden", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
           "-I", os.path.join(args.ref, "src"), "-o", so] + srcs
    print(" ".join(cmd))
    r = subprocess.run(cmd)
    if not args.keep:
        for p in srcs[:-1]:
            os.remove(p)
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
