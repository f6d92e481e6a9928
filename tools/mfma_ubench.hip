// v_mfma_i32_4x4x4_16b_i8 on gfx950 (MI355X): operand layout and issue cost, alone and interleaved with VALU work.
// The instruction multiplies, in each of 16 independent blocks (block b = lanes 4b..4b+3), a 4x4 i8 matrix A (lane 4b+i supplies
// row i as four packed bytes = the K dimension) with a 4x4 matrix B (lane 4b+j supplies column j) and adds C; this program prints
// which lane / register each D[i][j] lands in and how many SIMD cycles an instruction occupies.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_ubench tools/mfma_ubench.hip && ./mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const uint32_t* a, const uint32_t* b, int* out)
{
    v4i acc = { 0, 0, 0, 0 };
    acc = __builtin_amdgcn_mfma_i32_4x4x4i8(int(a[threadIdx.x]), int(b[threadIdx.x]), acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = acc[r];
}

constexpr int kIters = 4096;
#define MF(D) "v_mfma_i32_4x4x4_16b_i8 " D ", %16, %17, " D "\n"
#define MX(D) "v_max3_i32 " D ", " D ", %16, %17\n"
#define CC(D) "v_cmp_lt_i32 vcc, %16, " D "\n v_cndmask_b32 " D ", " D ", %17, vcc\n"

// MODE 0: 8 independent MFMAs per trip; 1: 8 MFMAs + 16 v_max3; 2: 16 v_max3 only; 3: 8 MFMAs + 8 cmp/cndmask pairs; 4: 8 cmp/cndmask pairs only; 5: one dependent MFMA chain
template<int MODE>
__global__ void rate_kernel(int* out, int sa, int sb)
{
    v4i m[8]; int r[8];
    for (int i = 0; i < 8; ++i) { m[i] = v4i{ i, sa, sb, int(threadIdx.x) }; r[i] = sa * i + int(threadIdx.x); }
    for (int it = 0; it < kIters; ++it)
    {
        if (MODE == 0)
            asm volatile(MF("%0") MF("%1") MF("%2") MF("%3") MF("%4") MF("%5") MF("%6") MF("%7")
                         : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]),
                           "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(sa), "v"(sb) : "vcc");
        else if (MODE == 1)
            asm volatile(MF("%0") MX("%8") MX("%9") MF("%1") MX("%10") MX("%11") MF("%2") MX("%12") MX("%13") MF("%3") MX("%14") MX("%15")
                         MF("%4") MX("%8") MX("%9") MF("%5") MX("%10") MX("%11") MF("%6") MX("%12") MX("%13") MF("%7") MX("%14") MX("%15")
                         : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]),
                           "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(sa), "v"(sb) : "vcc");
        else if (MODE == 2)
            asm volatile(MX("%8") MX("%9") MX("%10") MX("%11") MX("%12") MX("%13") MX("%14") MX("%15") MX("%8") MX("%9") MX("%10") MX("%11") MX("%12") MX("%13") MX("%14") MX("%15")
                         : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]),
                           "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(sa), "v"(sb) : "vcc");
        else if (MODE == 3)
            asm volatile(MF("%0") CC("%8") MF("%1") CC("%9") MF("%2") CC("%10") MF("%3") CC("%11") MF("%4") CC("%12") MF("%5") CC("%13") MF("%6") CC("%14") MF("%7") CC("%15")
                         : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]),
                           "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(sa), "v"(sb) : "vcc");
        else if (MODE == 4)
            asm volatile(CC("%8") CC("%9") CC("%10") CC("%11") CC("%12") CC("%13") CC("%14") CC("%15")
                         : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]),
                           "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(sa), "v"(sb) : "vcc");
        else
            asm volatile(MF("%0") MF("%0") MF("%0") MF("%0") MF("%0") MF("%0") MF("%0") MF("%0")
                         : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]),
                           "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(sa), "v"(sb) : "vcc");
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += m[i][0] + m[i][1] + m[i][2] + m[i][3] + r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<int MODE>
static double run(int wavesPerSimd, int* dOut, double ghz, int cus)
{
    const int blocks = cus * 4 * wavesPerSimd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, dOut, 3, 5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, dOut, 3, 5);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return double(ms) * 1e-3 * ghz * 1e9 / (double(kIters) * wavesPerSimd);       // SIMD cycles per loop trip per wave
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6; const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs, %.2f GHz\n", p.name, cus, ghz);
    // layout: A bytes of lane l = (l, l + 1, 0, 0) in K slots 0, 1; B bytes of lane l = (1, 0, 0, 0) -> D[i][j] = A[i][0] * B[0][j] = lane index of the row supplier
    std::vector<uint32_t> a(64), b(64); std::vector<int> d(256);
    for (int l = 0; l < 64; ++l) { a[l] = uint32_t(l & 0x7F) | (uint32_t((l + 1) & 0x7F) << 8); b[l] = 1u | (uint32_t(l & 3) << 8 & 0); }
    uint32_t *dA, *dB; int* dD; int* dOut;
    CHECK(hipMalloc(&dA, 256)); CHECK(hipMalloc(&dB, 256)); CHECK(hipMalloc(&dD, 1024)); CHECK(hipMalloc(&dOut, size_t(cus) * 4 * 8 * 64 * 4));
    CHECK(hipMemcpy(dA, a.data(), 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, b.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CHECK(hipMemcpy(d.data(), dD, 1024, hipMemcpyDeviceToHost));
    printf("## layout, B = e0 in every lane: D register r of lane l holds A[row][0] where the row's supplier lane is\n");
    for (int l = 0; l < 12; ++l) printf("lane %2d: r0 <- lane %d, r1 <- lane %d, r2 <- lane %d, r3 <- lane %d\n", l, d[l * 4], d[l * 4 + 1], d[l * 4 + 2], d[l * 4 + 3]);
    // second probe: A = e0 (byte0 = 1) in every lane, B byte0 = lane -> D[i][j] = B[0][j]: which column does lane l receive?
    for (int l = 0; l < 64; ++l) { a[l] = 1u; b[l] = uint32_t(l & 0x7F); }
    CHECK(hipMemcpy(dA, a.data(), 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, b.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CHECK(hipMemcpy(d.data(), dD, 1024, hipMemcpyDeviceToHost));
    printf("## layout, A = e0 in every lane: D register r of lane l holds B[0][col] where the column's supplier lane is\n");
    for (int l = 0; l < 12; ++l) printf("lane %2d: r0 <- lane %d, r1 <- lane %d, r2 <- lane %d, r3 <- lane %d\n", l, d[l * 4], d[l * 4 + 1], d[l * 4 + 2], d[l * 4 + 3]);
    // signedness probe: A byte0 = 0xFF (-1), B byte0 = 2 -> -2 if signed
    for (int l = 0; l < 64; ++l) { a[l] = 0xFFu; b[l] = 2u; }
    CHECK(hipMemcpy(dA, a.data(), 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, b.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CHECK(hipMemcpy(d.data(), dD, 1024, hipMemcpyDeviceToHost));
    printf("## 0xFF x 2 = %d (signed operands: -2)\n", d[0]);

    printf("## SIMD cycles per loop trip per wave (W waves per SIMD)\n| trip | W=1 | W=2 | W=4 | W=8 |\n|---|---|---|---|---|\n");
#define ROW(NAME, MODE) printf("| %s | %.1f | %.1f | %.1f | %.1f |\n", NAME, run<MODE>(1, dOut, ghz, cus), run<MODE>(2, dOut, ghz, cus), run<MODE>(4, dOut, ghz, cus), run<MODE>(8, dOut, ghz, cus));
    ROW("8 independent mfma_4x4x4_i8", 0)
    ROW("8 mfma, one dependent chain", 5)
    ROW("16 v_max3_i32", 2)
    ROW("8 mfma + 16 v_max3_i32 interleaved", 1)
    ROW("8 cmp+cndmask pairs", 4)
    ROW("8 mfma + 8 cmp+cndmask pairs interleaved", 3)
    return 0;
}
