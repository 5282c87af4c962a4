// tools/ubench/valu_ubench.hip -- issue cost of the integer VALU instructions the sampler leans on (gfx950),
// measured with inline asm so that the compiler cannot fold anything.  8 independent chains per lane,
// 8 waves per SIMD.  Prints SIMD-cycles per wave-instruction at the clock measured with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t a0, uint32_t b0, int iters) {
    uint32_t x0, x1, x2, x3, x4, x5, x6, x7;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    x0 = i + a0; x1 = i * 3 + a0; x2 = i * 5 + a0; x3 = i * 7 + a0; x4 = i * 9 + a0; x5 = i * 11 + a0; x6 = i * 13 + a0; x7 = i * 15 + a0;
    uint32_t c = b0 ^ i, d = a0 + i * 17;
    uint32_t cnt = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#define XOR(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define BITOP(n) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x##n) : "v"(c), "s"(b0));
#define ALIGN(n) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x##n));
#define ADD3(n) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x##n) : "v"(c), "v"(d));
#define LSHLOR(n) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x##n) : "v"(c));
#define MAX3(n) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(x##n) : "v"(c), "v"(d));
#define SUBCO(n) asm volatile("v_sub_co_u32 %0, vcc, %1, %2\n\ts_nop 1\n\tv_subbrev_co_u32 %3, vcc, 0, %3, vcc" : "=v"(x##n), "+v"(cnt) : "v"(c), "v"(d) : "vcc"); asm volatile("" : "+v"(cnt));
#define CMPCND(n) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x##n) : "v"(c), "v"(d) : "vcc");
#define SUB(n) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define MAD(n) { uint64_t p; asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, 0" : "=v"(p) : "v"(x##n), "s"(b0) : "s10", "s11"); x##n = (uint32_t)(p >> 32); }
#define SDWASUB(n) asm volatile("v_sub_co_u32_sdwa %0, vcc, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n\ts_nop 1\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "=v"(x##n), "+v"(cnt) : "v"(c), "v"(d) : "vcc"); asm volatile("" : "+v"(cnt));
#define PERM(n) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x##n) : "v"(c), "v"(d));
#define ANDV(n) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x##n));
#define LSHL(n) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x##n));
#define ADDC2(n) asm volatile("v_cmp_ne_u32 vcc, %1, %2\n\ts_nop 1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x##n) : "v"(c), "v"(d) : "vcc");
#define MIN3(n) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x##n) : "v"(c), "v"(d));
#define MINV(n) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define PAIRNONOP(n) asm volatile("v_sub_co_u32 %0, vcc, %1, %2\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "=v"(x##n), "+v"(cnt) : "v"(c), "v"(d) : "vcc"); asm volatile("" : "+v"(cnt));
#define PAIRFILL(n) asm volatile("v_sub_co_u32 %0, vcc, %1, %2\n\tv_xor_b32 %4, %4, %1\n\tv_xor_b32 %5, %5, %2\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "=v"(x##n), "+v"(cnt), "+v"(c) : "v"(c), "v"(d), "v"(e1), "v"(e2) : "vcc");
#define NOPONLY(n) asm volatile("s_nop 1");
#define XORNOP(n) asm volatile("v_xor_b32 %0, %0, %1\n\ts_nop 1" : "+v"(x##n) : "v"(c));
#define MULLO(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define MULHI(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define MULU24(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define MADU24(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x##n) : "v"(c), "v"(d));
#define PKSUBC(n) asm volatile("v_pk_sub_u16 %0, %0, %1 op_sel_hi:[1,1] clamp" : "+v"(x##n) : "v"(c));
#define PKSUBSEL(n) asm volatile("v_pk_sub_u16 %0, %0, %1 op_sel:[1,0] op_sel_hi:[1,1] clamp" : "+v"(x##n) : "v"(c));
#define PKMIN(n) asm volatile("v_pk_min_u16 %0, %0, 2 op_sel_hi:[1,0]" : "+v"(x##n));
#define PKMINV(n) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define PKADD(n) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define PKLSHR(n) asm volatile("v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]" : "+v"(x##n));
#define ANDOR(n) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x##n) : "v"(c), "v"(d));
#define BFE(n) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(x##n));
#define BCNT(n) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define ORLIT(n) asm volatile("v_or_b32 %0, 0x0c000c00, %0" : "+v"(x##n));
#define ADDV(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x##n) : "v"(c));
#define LSHR(n) asm volatile("v_lshrrev_b32 %0, 15, %0" : "+v"(x##n));
#define ADDDPP(n) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(x##n));
#define DOT2(n) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(x##n) : "v"(c), "v"(d));
#define ADDSDWA(n) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(x##n) : "v"(c));
#define SWAR4(n) asm volatile("v_sub_u32 %0, %1, %0\n\tv_lshrrev_b32 %0, 15, %0\n\tv_and_b32 %0, 0x00010001, %0\n\tv_add_u32 %0, %0, %2" : "+v"(x##n) : "v"(c), "v"(d));
#define PK3(n) asm volatile("v_pk_sub_u16 %0, %1, %0 clamp\n\tv_pk_min_u16 %0, %0, 2 op_sel_hi:[1,0]\n\tv_pk_add_u16 %0, %0, %2" : "+v"(x##n) : "v"(c), "v"(d));
#define FFBL(n) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x##n));
#define MOVV(n) asm volatile("v_mov_b32 %0, %1" : "=v"(x##n) : "v"(c));
#define CNDM(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x##n) : "v"(c) : "vcc");
#define ADDF64(n) { double q = __longlong_as_double(((long long)x##n << 32) | x##n); asm volatile("v_add_f64 %0, %0, %1" : "+v"(q) : "v"(q)); x##n = (uint32_t)__double_as_longlong(q); }
            if (KIND == 30) { REP8(PKSUBC) }
            if (KIND == 31) { REP8(PKSUBSEL) }
            if (KIND == 32) { REP8(PKMIN) }
            if (KIND == 33) { REP8(PKMINV) }
            if (KIND == 34) { REP8(PKADD) }
            if (KIND == 35) { REP8(PKLSHR) }
            if (KIND == 36) { REP8(ANDOR) }
            if (KIND == 37) { REP8(BFE) }
            if (KIND == 38) { REP8(BCNT) }
            if (KIND == 39) { REP8(ORLIT) }
            if (KIND == 40) { REP8(ADDV) }
            if (KIND == 41) { REP8(LSHR) }
            if (KIND == 42) { REP8(ADDDPP) }
            if (KIND == 43) { REP8(DOT2) }
            if (KIND == 44) { REP8(ADDSDWA) }
            if (KIND == 45) { REP8(SWAR4) }
            if (KIND == 46) { REP8(PK3) }
            if (KIND == 47) { REP8(FFBL) }
            if (KIND == 48) { REP8(MOVV) }
            if (KIND == 49) { REP8(CNDM) }
            if (KIND == 21) { REP8(MULLO) }
            if (KIND == 22) { REP8(MULU24) }
            if (KIND == 24) { REP8(MULHI) }
            if (KIND == 23) { REP8(MADU24) }
            if (KIND == 17) { REP8(PAIRNONOP) }
            if (KIND == 19) { REP8(NOPONLY) }
            if (KIND == 20) { REP8(XORNOP) }
            if (KIND == 10) { REP8(SDWASUB) }
            if (KIND == 11) { REP8(PERM) }
            if (KIND == 12) { REP8(ANDV) }
            if (KIND == 13) { REP8(LSHL) }
            if (KIND == 14) { REP8(ADDC2) }
            if (KIND == 15) { REP8(MIN3) }
            if (KIND == 16) { REP8(MINV) }
            if (KIND == 0) { REP8(XOR) }
            if (KIND == 1) { REP8(BITOP) }
            if (KIND == 2) { REP8(ALIGN) }
            if (KIND == 3) { REP8(ADD3) }
            if (KIND == 4) { REP8(MAD) }
            if (KIND == 5) { REP8(SUBCO) }
            if (KIND == 6) { REP8(LSHLOR) }
            if (KIND == 7) { REP8(MAX3) }
            if (KIND == 8) { REP8(CMPCND) }
            if (KIND == 9) { REP8(SUB) }
        }
    }
    out[i] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + cnt;
}

template <int KIND>
void run(const char *name, int instr_per_op) {
    const int blocks = 256 * 8, iters = 256;
    uint32_t *d;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(d, 1, 2, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k<KIND><<<blocks, 256>>>(d, 1, 2, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr = 5.0 * blocks * 4.0 * iters * 8 * 8 * instr_per_op;
    const double cyc = (ms * 1e-3) * 2.4e9 * 1024.0 / wave_instr;
    printf("%-34s %7.3f ms  %.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, ms / 5, cyc);
    hipFree(d);
}

int main() {
    run<0>("v_xor_b32 (VOP2)", 1);
    run<9>("v_sub_u32 (VOP2)", 1);
    run<1>("v_bitop3_b32 (VOP3, 1 sgpr)", 1);
    run<2>("v_alignbit_b32", 1);
    run<3>("v_add3_u32", 1);
    run<6>("v_lshl_or_b32", 1);
    run<7>("v_max3_u32", 1);
    run<4>("v_mad_u64_u32", 1);
    run<5>("v_sub_co + s_nop 1 + v_subbrev_co", 2);
    run<8>("v_cmp_lt + s_nop 1 + v_cndmask", 2);
    run<17>("v_sub_co + v_addc (no s_nop; timing only)", 2);
    run<19>("s_nop 1 alone (per s_nop)", 1);
    run<20>("v_xor + s_nop 1 (per v_xor)", 1);
    run<10>("v_sub_co_sdwa + s_nop 1 + v_addc", 2);
    run<14>("v_cmp_ne + s_nop 1 + v_addc", 2);
    run<11>("v_perm_b32", 1);
    run<21>("v_mul_lo_u32", 1);
    run<24>("v_mul_hi_u32", 1);
    run<22>("v_mul_u32_u24 (VOP2)", 1);
    run<23>("v_mad_u32_u24", 1);
    run<15>("v_min3_u32", 1);
    run<16>("v_min_u32 (VOP2)", 1);
    run<30>("v_pk_sub_u16 clamp", 1);
    run<31>("v_pk_sub_u16 clamp op_sel", 1);
    run<32>("v_pk_min_u16 inline const", 1);
    run<33>("v_pk_min_u16 vgpr", 1);
    run<34>("v_pk_add_u16", 1);
    run<35>("v_pk_lshrrev_b16", 1);
    run<36>("v_and_or_b32", 1);
    run<37>("v_bfe_u32", 1);
    run<38>("v_bcnt_u32_b32", 1);
    run<39>("v_or_b32 literal (VOP2)", 1);
    run<40>("v_add_u32 (VOP2)", 1);
    run<41>("v_lshrrev_b32 (VOP2)", 1);
    run<42>("v_add_u32_dpp row_shr:1", 1);
    run<43>("v_dot2_u32_u16", 1);
    run<44>("v_add_u32_sdwa", 1);
    run<45>("SWAR4: sub,lshr,and,add (per instr)", 4);
    run<46>("PK3: pk_sub clamp,pk_min,pk_add (per instr)", 3);
    run<47>("v_ffbl_b32 (VOP1)", 1);
    run<48>("v_mov_b32 (VOP1)", 1);
    run<49>("v_cndmask_b32 vcc (VOP2)", 1);
    run<12>("v_and_b32 literal (VOP2)", 1);
    run<13>("v_lshlrev_b32 (VOP2)", 1);
    return 0;
}
