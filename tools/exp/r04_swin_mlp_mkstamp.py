"""Writes a stamped copy of edtr_amd/csrc/swin.hip (s_memtime stamps around the periods of swin_mlp_kernel, read back by
r04_swin_mlp_stamps.py through a diagnostic export edtr_mlp_dbg): python r04_swin_mlp_mkstamp.py <swin.hip> <out.hip>; build the
result into a library of its own and point EDTR_AMD_LIB at it.  Diagnostic builds only — the shipped kernel carries no stamp."""
import sys
src, dst = sys.argv[1], sys.argv[2]
s=open(src).read()
s=s.replace("constexpr int MLP_CT = 6, MLP_HT = 12;","__device__ unsigned long long g_mlp_dbg[4 * 64];\n#define STAMP(k) do { if (lane == 0 && (wave == 0 || wave == 4) && (blockIdx.x == 0 || blockIdx.x == 200)) g_mlp_dbg[((blockIdx.x ? 2 : 0) + (wave >> 2)) * 64 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)\nconstexpr int MLP_CT = 6, MLP_HT = 12;",1)
s=s.replace("    const int tok_base = blockIdx.x * MLP_TOKENS;\n","    const int tok_base = blockIdx.x * MLP_TOKENS;\n    STAMP(0);\n",1)
s=s.replace("    stage_unit(0);\n    stage_unit(1);\n","    stage_unit(0);\n    stage_unit(1);\n    STAMP(1);\n",1)
s=s.replace("        __syncthreads();\n        if (t + 2 < PERIODS) stage_unit(t + 2);","        STAMP(2 + 3 * t);\n        __syncthreads();\n        STAMP(3 + 3 * t);\n        if (t + 2 < PERIODS) stage_unit(t + 2);",1)
s=s.replace("                if (s == 0) hb0 = pack8<T>(g); else hb1 = pack8<T>(g);\n            }\n        }\n    }\n","                if (s == 0) hb0 = pack8<T>(g); else hb1 = pack8<T>(g);\n            }\n        }\n        STAMP(4 + 3 * t);\n    }\n",1)
s=s.replace("    // ---- the two hidden halves meet: wave (t4, hg) keeps output tiles 3 hg .. 3 hg + 2 and hands over the other three\n    __syncthreads();","    STAMP(50);\n    __syncthreads();\n    STAMP(51);",1)
s=s.replace("    if (hg == 0) hand_over(std::integral_constant<int, 0>{}); else hand_over(std::integral_constant<int, 1>{});\n    __syncthreads();","    if (hg == 0) hand_over(std::integral_constant<int, 0>{}); else hand_over(std::integral_constant<int, 1>{});\n    STAMP(52);\n    __syncthreads();\n    STAMP(53);",1)
s=s.replace("    if (hg == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 1>{});\n    __syncthreads();","    if (hg == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 1>{});\n    STAMP(54);\n    __syncthreads();\n    STAMP(55);",1)
s=s.replace("            if (tok_base + r < p.rows) stg16(og + (int64_t)(tok_base + r) * p.ldo + c * 8, v);\n        }\n    }\n}","            if (tok_base + r < p.rows) stg16(og + (int64_t)(tok_base + r) * p.ldo + c * 8, v);\n        }\n        STAMP(56);\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        STAMP(57);\n    }\n}",1)
s=s.replace('extern "C" int edtr_swin_mlp(','extern "C" int edtr_mlp_dbg(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_mlp_dbg), sizeof(unsigned long long) * 256); }\n\nextern "C" int edtr_swin_mlp(',1)
assert s.count("STAMP(") >= 12, s.count("STAMP(")
open(dst,'w').write(s)
