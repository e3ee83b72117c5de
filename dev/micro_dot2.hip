// Is r = x - bf16(x) exact through v_dot2c_f32_bf16 (packed {h0, h1} . {-1, 0} + x0)?  Compares the three planes of split3's
// shift/mask/subtract form with the dot2 form bit for bit over random and edge-case fp32 inputs.
// build: hipcc --offload-arch=gfx950 -O3 -o dev/micro_dot2 dev/micro_dot2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__global__ void k_old(const float* x, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float x0 = x[2 * i], x1 = x[2 * i + 1];
    unsigned h = cvt_pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    unsigned m = cvt_pk_bf16(r0, r1);
    float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    unsigned l = cvt_pk_bf16(s0, s1);
    out[3 * i] = h; out[3 * i + 1] = m; out[3 * i + 2] = l;
}
template <bool SGPR>
__global__ void k_new(const float* x, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float x0 = x[2 * i], x1 = x[2 * i + 1];
    unsigned c0u = 0x0000BF80u, c1u = 0xBF800000u;
    if (SGPR) { asm volatile("" : "+s"(c0u)); asm volatile("" : "+s"(c1u)); }
    const bf16x2 c0 = __builtin_bit_cast(bf16x2, c0u), c1 = __builtin_bit_cast(bf16x2, c1u);
    unsigned h = cvt_pk_bf16(x0, x1);
    float r0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), c0, x0, false);
    float r1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), c1, x1, false);
    unsigned m = cvt_pk_bf16(r0, r1);
    float s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, m), c0, r0, false);
    float s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, m), c1, r1, false);
    unsigned l = cvt_pk_bf16(s0, s1);
    out[3 * i] = h; out[3 * i + 1] = m; out[3 * i + 2] = l;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n);
    uint64_t st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        uint32_t b = (uint32_t)(st >> 16);
        if (i % 4 == 0) { float f; memcpy(&f, &b, 4); if (!std::isfinite(f)) f = 1.0f; hx[i] = f; }           // any finite bit pattern
        else if (i % 4 == 1) hx[i] = (float)((double)(int32_t)b / 2147483648.0);                                // (-1, 1)
        else if (i % 4 == 2) hx[i] = (float)((double)(int32_t)b / 2147483648.0) * 1e-30f;                         // small
        else hx[i] = (float)((double)(int32_t)b / 2147483648.0) * 3e4f;                                           // data-sized
    }
    hx[0] = 0.f; hx[1] = -0.f; hx[2] = 1.f; hx[3] = -1.f; hx[4] = 1.00390625f; hx[5] = 1e-40f; hx[6] = 3.3e38f; hx[7] = -1.17549435e-38f;
    float* dx; unsigned *d0, *d1, *d2;
    hipMalloc(&dx, 4 * n); hipMalloc(&d0, 6 * n); hipMalloc(&d1, 6 * n); hipMalloc(&d2, 6 * n);
    hipMemcpy(dx, hx.data(), 4 * n, hipMemcpyHostToDevice);
    k_old<<<n / 2 / 256, 256>>>(dx, d0, n);
    k_new<false><<<n / 2 / 256, 256>>>(dx, d1, n);
    k_new<true><<<n / 2 / 256, 256>>>(dx, d2, n);
    std::vector<unsigned> o0(3 * n / 2), o1(3 * n / 2), o2(3 * n / 2);
    hipMemcpy(o0.data(), d0, 6 * n, hipMemcpyDeviceToHost);
    hipMemcpy(o1.data(), d1, 6 * n, hipMemcpyDeviceToHost);
    hipMemcpy(o2.data(), d2, 6 * n, hipMemcpyDeviceToHost);
    long bad1 = 0, bad2 = 0, bad1n = 0, bad2n = 0;   // *n: mismatches on inputs of normal size (|x| in [1e-30, 1e30])
    for (int i = 0; i < n / 2; ++i) {
        const bool normal = std::fabs(hx[2 * i]) > 1e-25f && std::fabs(hx[2 * i]) < 1e30f && std::fabs(hx[2 * i + 1]) > 1e-25f && std::fabs(hx[2 * i + 1]) < 1e30f;
        bool b1 = false, b2 = false;
        for (int p = 0; p < 3; ++p) { b1 |= o0[3 * i + p] != o1[3 * i + p]; b2 |= o0[3 * i + p] != o2[3 * i + p]; }
        bad1 += b1; bad2 += b2; bad1n += b1 && normal; bad2n += b2 && normal;
        if ((b1 || b2) && bad1 + bad2 < 12)
            printf("pair %d x = %a %a: old %08x %08x %08x, inline %08x %08x %08x, sgpr %08x %08x %08x\n", i, hx[2 * i], hx[2 * i + 1], o0[3 * i], o0[3 * i + 1],
                   o0[3 * i + 2], o1[3 * i], o1[3 * i + 1], o1[3 * i + 2], o2[3 * i], o2[3 * i + 1], o2[3 * i + 2]);
    }
    printf("pairs %d: mismatching (inline constants) %ld, of normal size %ld; (constants in SGPRs) %ld, of normal size %ld\n", n / 2, bad1, bad1n, bad2, bad2n);
    return 0;
}
