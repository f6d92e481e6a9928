// VALU issue-rate micro-benchmark for gfx950 (MI355X): how many SIMD cycles one wave64 instruction of each kind
// occupies, measured with every SIMD of the chip holding W waves that run long unrolled streams of that instruction
// over 8 independent register chains (throughput) or one chain (dependent latency). The BC7 / BC6H search kernels
// are VALU-bound integer code, so their roofline is "lane-operations per second", weighted by the issue cost of
// the instructions they actually use (profiles/r02_valu_rates.md is this program's output).
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_ubench tools/valu_ubench.hip && ./valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kUnroll = 64;      // instructions per loop trip
constexpr int kIters = 2048;     // loop trips

// One instruction on chain registers (d = destination/accumulator, a, b = sources). %0..%7 are the 8 chain registers,
// %8, %9 two loop-invariant source registers.
#define OP8(INSN) \
    INSN("%0") INSN("%1") INSN("%2") INSN("%3") INSN("%4") INSN("%5") INSN("%6") INSN("%7")
#define OP1(INSN) \
    INSN("%0") INSN("%0") INSN("%0") INSN("%0") INSN("%0") INSN("%0") INSN("%0") INSN("%0")

#define BODY(CH, INSN) \
    for (int it = 0; it < kIters; ++it) { \
        asm volatile(CH(INSN) CH(INSN) CH(INSN) CH(INSN) CH(INSN) CH(INSN) CH(INSN) CH(INSN) \
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                     : "v"(sa), "v"(sb) : "vcc"); }

// 64-bit chains
#define BODY64(CH, INSN) \
    for (int it = 0; it < kIters; ++it) { \
        asm volatile(CH(INSN) CH(INSN) CH(INSN) CH(INSN) CH(INSN) CH(INSN) CH(INSN) CH(INSN) \
                     : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) \
                     : "v"(sa), "v"(sb) : "vcc"); }

#define I_DOT4U(D)   "v_dot4_u32_u8 " D ", %8, %9, " D "\n"
#define I_DOT4I(D)   "v_dot4_i32_i8 " D ", %8, %9, " D "\n"
#define I_DOT2(D)    "v_dot2_u32_u16 " D ", %8, %9, " D "\n"
#define I_DOT8(D)    "v_dot8_u32_u4 " D ", %8, %9, " D "\n"
#define I_ADD(D)     "v_add_u32 " D ", %8, " D "\n"
#define I_SUB(D)     "v_sub_u32 " D ", " D ", %8\n"
#define I_AND(D)     "v_and_b32 " D ", %8, " D "\n"
#define I_OR3(D)     "v_or3_b32 " D ", " D ", %8, %9\n"
#define I_ADD3(D)    "v_add3_u32 " D ", " D ", %8, %9\n"
#define I_LSHLADD(D) "v_lshl_add_u32 " D ", " D ", 1, %8\n"
#define I_LSHL(D)    "v_lshlrev_b32 " D ", 1, " D "\n"
#define I_LSHRV(D)   "v_lshrrev_b32 " D ", %8, " D "\n"
#define I_BFE(D)     "v_bfe_u32 " D ", " D ", 3, 8\n"
#define I_BFI(D)     "v_bfi_b32 " D ", %8, %9, " D "\n"
#define I_PERM(D)    "v_perm_b32 " D ", " D ", %8, %9\n"
#define I_MAXI(D)    "v_max_i32 " D ", %8, " D "\n"
#define I_MAX3(D)    "v_max3_i32 " D ", " D ", %8, %9\n"
#define I_MED3(D)    "v_med3_i32 " D ", " D ", %8, %9\n"
#define I_MULLO(D)   "v_mul_lo_u32 " D ", " D ", %8\n"
#define I_MULU24(D)  "v_mul_u32_u24 " D ", " D ", %8\n"
#define I_MADU24(D)  "v_mad_u32_u24 " D ", " D ", %8, %9\n"
#define I_MADI24(D)  "v_mad_i32_i24 " D ", " D ", %8, %9\n"
#define I_SAD(D)     "v_sad_u8 " D ", %8, %9, " D "\n"
#define I_CMP(D)     "v_cmp_lt_i32 vcc, %8, " D "\n"
#define I_CNDVCC(D)  "v_cndmask_b32 " D ", " D ", %8, vcc\n"
#define I_CMPCND(D)  "v_cmp_lt_i32 vcc, %8, " D "\n v_cndmask_b32 " D ", " D ", %9, vcc\n"
#define I_CMPCND_S(D) "v_cmp_lt_i32 s[20:21], %8, " D "\n v_cndmask_b32 " D ", " D ", %9, s[20:21]\n"
#define I_FMA(D)     "v_fma_f32 " D ", %8, %9, " D "\n"
#define I_FMUL(D)    "v_mul_f32 " D ", %8, " D "\n"
#define I_FADD(D)    "v_add_f32 " D ", %8, " D "\n"
#define I_FMAX(D)    "v_max_f32 " D ", %8, " D "\n"
#define I_RCP(D)     "v_rcp_f32 " D ", " D "\n"
#define I_SQRT(D)    "v_sqrt_f32 " D ", " D "\n"
#define I_CVTFU(D)   "v_cvt_f32_u32 " D ", " D "\n"
#define I_CVTUF(D)   "v_cvt_u32_f32 " D ", " D "\n"
#define I_CVTF16(D)  "v_cvt_f16_f32 " D ", " D "\n"
#define I_MOV(D)     "v_mov_b32 " D ", %8\n"
#define I_PKMULLO(D) "v_pk_mul_lo_u16 " D ", " D ", %8\n"
#define I_PKMAD(D)   "v_pk_mad_u16 " D ", " D ", %8, %9\n"
#define I_PKADD(D)   "v_pk_add_u16 " D ", " D ", %8\n"
#define I_PKFMA16(D) "v_pk_fma_f16 " D ", %8, %9, " D "\n"
#define I_PKMAXI16(D) "v_pk_max_i16 " D ", " D ", %8\n"
#define I_MBCNT(D)   "v_mbcnt_lo_u32_b32 " D ", %8, " D "\n"
#define I_DPP(D)     "v_add_u32_dpp " D ", " D ", " D " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_SDWA(D)    "v_add_u32_sdwa " D ", " D ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
// 64-bit
#define I_LSHL64(D)  "v_lshlrev_b64 " D ", 1, " D "\n"
#define I_LSHR64(D)  "v_lshrrev_b64 " D ", %8, " D "\n"
#define I_PKFMA32(D) "v_pk_fma_f32 " D ", " D ", " D ", " D "\n"
#define I_FMA64(D)   "v_fma_f64 " D ", " D ", " D ", " D "\n"
#define I_MAD64(D)   "v_mad_u64_u32 " D ", vcc, %8, %9, " D "\n"

enum Op { DOT4U, DOT4I, DOT2, DOT8, ADD, SUB, AND, OR3, ADD3, LSHLADD, LSHL, LSHRV, BFE, BFI, PERM, MAXI, MAX3, MED3, MULLO, MULU24, MADU24, MADI24, SAD,
          CMP, CNDVCC, CMPCND, CMPCND_S, FMA, FMUL, FADD, FMAX, RCP, SQRT, CVTFU, CVTUF, CVTF16, MOV, PKMULLO, PKMAD, PKADD, PKFMA16, PKMAXI16, MBCNT, DPP, SDWA,
          LSHL64, LSHR64, PKFMA32, FMA64, MAD64, DSREAD, DSREAD64, DSREAD128, NUM_OPS };

struct OpInfo { const char* name; int instsPerSlot; };
static const OpInfo kOps[NUM_OPS] = {
    { "v_dot4_u32_u8", 1 }, { "v_dot4_i32_i8", 1 }, { "v_dot2_u32_u16", 1 }, { "v_dot8_u32_u4", 1 }, { "v_add_u32", 1 }, { "v_sub_u32", 1 }, { "v_and_b32", 1 },
    { "v_or3_b32", 1 }, { "v_add3_u32", 1 }, { "v_lshl_add_u32", 1 }, { "v_lshlrev_b32 (imm)", 1 }, { "v_lshrrev_b32 (reg)", 1 }, { "v_bfe_u32", 1 }, { "v_bfi_b32", 1 },
    { "v_perm_b32", 1 }, { "v_max_i32", 1 }, { "v_max3_i32", 1 }, { "v_med3_i32", 1 }, { "v_mul_lo_u32", 1 }, { "v_mul_u32_u24", 1 }, { "v_mad_u32_u24", 1 },
    { "v_mad_i32_i24", 1 }, { "v_sad_u8", 1 }, { "v_cmp_lt_i32 -> vcc", 1 }, { "v_cndmask_b32 (vcc)", 1 }, { "v_cmp_lt_i32 vcc + v_cndmask_b32 (pair)", 2 },
    { "v_cmp_lt_i32 sgpr + v_cndmask_b32 (pair)", 2 }, { "v_fma_f32", 1 }, { "v_mul_f32", 1 }, { "v_add_f32", 1 }, { "v_max_f32", 1 }, { "v_rcp_f32", 1 }, { "v_sqrt_f32", 1 },
    { "v_cvt_f32_u32", 1 }, { "v_cvt_u32_f32", 1 }, { "v_cvt_f16_f32", 1 }, { "v_mov_b32", 1 }, { "v_pk_mul_lo_u16", 1 }, { "v_pk_mad_u16", 1 }, { "v_pk_add_u16", 1 },
    { "v_pk_fma_f16", 1 }, { "v_pk_max_i16", 1 }, { "v_mbcnt_lo_u32_b32", 1 }, { "v_add_u32_dpp row_shr:1", 1 }, { "v_add_u32_sdwa (byte select)", 1 },
    { "v_lshlrev_b64 (imm)", 1 }, { "v_lshrrev_b64 (reg)", 1 }, { "v_pk_fma_f32", 1 }, { "v_fma_f64", 1 }, { "v_mad_u64_u32", 1 },
    { "ds_read_b32 (conflict-free column)", 1 }, { "ds_read_b64", 1 }, { "ds_read_b128", 1 },
};

template<int OP, bool DEP>
__global__ void __launch_bounds__(64) ubench(uint32_t* out, unsigned long long* cycles, uint32_t sa, uint32_t sb)
{
    __shared__ uint32_t lds[64 * 8];
    uint32_t r[8];
    uint64_t q[8];
    for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x * 8u + i + sa; q[i] = (uint64_t(r[i]) << 20) | sb; }
    lds[threadIdx.x] = sa; lds[threadIdx.x + 64] = sb;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if constexpr (OP >= DSREAD)
    {
        // LDS reads: address chain-independent, results accumulated so the loads are kept
        uint32_t acc = 0;
        const uint32_t addr = threadIdx.x * ((OP == DSREAD) ? 4u : (OP == DSREAD64) ? 8u : 16u);
        for (int it = 0; it < kIters; ++it)
        {
            if (OP == DSREAD)
            {
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(0));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k) acc ^= v[k];
            }
            else if (OP == DSREAD64)
            {
                uint64_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("ds_read_b64 %0, %1" : "=v"(v[k]) : "v"(addr));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k) acc ^= uint32_t(v[k]);
            }
            else
            {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                u4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(addr & 2047u));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k) acc ^= v[k].x;
            }
        }
        r[0] = acc;
    }
#define CASE(ID, INSN) else if constexpr (OP == ID) { if (DEP) { BODY(OP1, INSN) } else { BODY(OP8, INSN) } }
#define CASE64(ID, INSN) else if constexpr (OP == ID) { if (DEP) { BODY64(OP1, INSN) } else { BODY64(OP8, INSN) } }
    CASE(DOT4U, I_DOT4U) CASE(DOT4I, I_DOT4I) CASE(DOT2, I_DOT2) CASE(DOT8, I_DOT8) CASE(ADD, I_ADD) CASE(SUB, I_SUB) CASE(AND, I_AND) CASE(OR3, I_OR3)
    CASE(ADD3, I_ADD3) CASE(LSHLADD, I_LSHLADD) CASE(LSHL, I_LSHL) CASE(LSHRV, I_LSHRV) CASE(BFE, I_BFE) CASE(BFI, I_BFI) CASE(PERM, I_PERM) CASE(MAXI, I_MAXI)
    CASE(MAX3, I_MAX3) CASE(MED3, I_MED3) CASE(MULLO, I_MULLO) CASE(MULU24, I_MULU24) CASE(MADU24, I_MADU24) CASE(MADI24, I_MADI24) CASE(SAD, I_SAD)
    CASE(CMP, I_CMP) CASE(CNDVCC, I_CNDVCC) CASE(CMPCND, I_CMPCND) CASE(CMPCND_S, I_CMPCND_S) CASE(FMA, I_FMA) CASE(FMUL, I_FMUL) CASE(FADD, I_FADD) CASE(FMAX, I_FMAX)
    CASE(RCP, I_RCP) CASE(SQRT, I_SQRT) CASE(CVTFU, I_CVTFU) CASE(CVTUF, I_CVTUF) CASE(CVTF16, I_CVTF16) CASE(MOV, I_MOV) CASE(PKMULLO, I_PKMULLO) CASE(PKMAD, I_PKMAD)
    CASE(PKADD, I_PKADD) CASE(PKFMA16, I_PKFMA16) CASE(PKMAXI16, I_PKMAXI16) CASE(MBCNT, I_MBCNT) CASE(DPP, I_DPP) CASE(SDWA, I_SDWA)
    CASE64(LSHL64, I_LSHL64) CASE64(LSHR64, I_LSHR64) CASE64(PKFMA32, I_PKFMA32) CASE64(FMA64, I_FMA64) CASE64(MAD64, I_MAD64)
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= r[i] ^ uint32_t(q[i]) ^ uint32_t(q[i] >> 32);
    if (s == 0x12345678u) out[0] = s + lds[(threadIdx.x * 7) & 127];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

typedef void (*KernelFn)(uint32_t*, unsigned long long*, uint32_t, uint32_t);
template<int OP> struct Table { static void fill(KernelFn (*t)[2]) { t[OP][0] = ubench<OP, false>; t[OP][1] = ubench<OP, true>; Table<OP - 1>::fill(t); } };
template<> struct Table<-1> { static void fill(KernelFn (*)[2]) {} };

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int clockKHz = 0;
    CHECK(hipDeviceGetAttribute(&clockKHz, hipDeviceAttributeClockRate, 0));
    int wallKHz = 0;
    (void)hipDeviceGetAttribute(&wallKHz, hipDeviceAttributeWallClockRate, 0);
    printf("# VALU issue rates on %s (%s): %d CUs, max clock %d MHz, s_memtime/readcyclecounter rate %d kHz\n\n", prop.name, prop.gcnArchName, cus, clockKHz / 1000, wallKHz);
    static KernelFn table[NUM_OPS][2];
    Table<NUM_OPS - 1>::fill(table);
    uint32_t* out; unsigned long long* cyc;
    const int maxBlocks = cus * 4 * 8;
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMalloc(&cyc, maxBlocks * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("Every SIMD holds W waves (grid = CUs x 4 x W workgroups of 64 threads), each wave issues %d x %d instructions of one kind.\n"
           "`cyc/inst/SIMD` = kernel wall time x clock / (W x instructions per wave): SIMD cycles one wave64 instruction occupies when the\n"
           "SIMD always has another wave to issue from (throughput), at the clock derived from the in-kernel cycle counter.\n"
           "`dep` = one wave per SIMD, one dependent chain: issue-to-issue latency of back-to-back dependent instructions.\n\n", kIters, kUnroll);
    printf("| instruction | cyc/inst/SIMD W=8 | W=4 | W=2 | W=1 (8 indep. chains) | dep chain, W=1 | dep chain, W=8 | lane-ops/s at W=8 (T) |\n|---|---|---|---|---|---|---|---|\n");
    for (int op = 0; op < NUM_OPS; ++op)
    {
        const double insts = double(kIters) * kUnroll * kOps[op].instsPerSlot * ((op >= DSREAD) ? 8.0 / kUnroll : 1.0);
        double res[6]; double lane = 0;
        const int Ws[6] = { 8, 4, 2, 1, 1, 8 };
        for (int c = 0; c < 6; ++c)
        {
            const int W = Ws[c]; const bool dep = c >= 4;
            const int blocks = cus * 4 * W;
            KernelFn fn = table[op][dep ? 1 : 0];
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(64), 0, 0, out, cyc, 3u, 5u);        // warm-up
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(64), 0, 0, out, cyc, 3u, 5u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(blocks);
            CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double mean = 0; for (auto v : h) mean += double(v); mean /= blocks;
            // in-kernel counter ticks at wallKHz (constant) on gfx9: convert the mean in-kernel duration to seconds, the kernel's wall time
            // to the same; cycles at the nominal max clock
            const double secs = ms * 1e-3;
            const double cycPerInst = secs * (double(clockKHz) * 1e3) / (double(W) * insts);
            res[c] = cycPerInst;
            if (c == 0) lane = double(blocks) * insts * 64.0 / secs / 1e12;
            (void)mean;
        }
        printf("| `%s` | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.1f |\n", kOps[op].name, res[0], res[1], res[2], res[3], res[4], res[5], lane);
        fflush(stdout);
    }
    printf("\nPeak for a 1-cycle... see profiles/r02_valu_rates.md for the reading. Wave64 VALU peak at 2 cyc/inst: %d CUs x 4 SIMDs x 32 lanes x %.1f GHz = %.1f T lane-ops/s.\n",
           cus, clockKHz / 1e6, cus * 4 * 32.0 * clockKHz / 1e9 / 1e3);
    return 0;
}
