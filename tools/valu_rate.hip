// micro-benchmark: issue rate of VALU instructions on gfx950 -- packed vs scalar f32, VOP2 vs VOP3 vs literal forms,
// the 32-bit integer multiplies and the transcendental unit.  8 independent destinations, 128 instructions per loop
// trip, 1 / 2 / 4 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
// the instruction is  A <reg> B  with <reg> one of the 8 chain registers (%0..%7); %8 / %9 are loop-invariant inputs
#define X8(A, B) A "%0" B "\n" A "%1" B "\n" A "%2" B "\n" A "%3" B "\n" A "%4" B "\n" A "%5" B "\n" A "%6" B "\n" A "%7" B
#define F32(A, B) REP16(asm volatile(X8(A, B) : "+v"(a0.x),"+v"(a1.x),"+v"(a2.x),"+v"(a3.x),"+v"(a4.x),"+v"(a5.x),"+v"(a6.x),"+v"(a7.x) : "v"(b.x), "v"(b.y) : "vcc", "s10", "s11");)
#define PK(A, B)  REP16(asm volatile(X8(A, B) : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b), "v"(b));)
#define U64(A, B) REP16(asm volatile(X8(A, B) : "+v"(q0),"+v"(q1),"+v"(q2),"+v"(q3),"+v"(q4),"+v"(q5),"+v"(q6),"+v"(q7) : "v"(b.x), "v"(b.y) : "vcc");)
template <int MODE>
__global__ void k(float* out, int iters)
{
    v2f a0={1.f,2.f},a1={3.f,4.f},a2={5.f,6.f},a3={7.f,8.f},a4={1.5f,2.5f},a5={3.5f,4.5f},a6={5.5f,6.5f},a7={7.5f,8.5f};
    unsigned long long q0=1,q1=2,q2=3,q3=4,q4=5,q5=6,q6=7,q7=8;
    v2f b={1e-9f*threadIdx.x,1e-9f};
    for (int i=0;i<iters;i++){
        if constexpr (MODE==0)  { PK("v_pk_add_f32 ", ",%8,%8") }
        if constexpr (MODE==1)  { F32("v_add_f32 ", ",%8,%9") }
        if constexpr (MODE==2)  { PK("v_pk_fma_f32 ", ",%8,%8,%8") }
        if constexpr (MODE==3)  { F32("v_fma_f32 ", ",%8,%9,%9") }
        if constexpr (MODE==4)  { PK("v_pk_add_f32 ", ",%8,%8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]") }
        if constexpr (MODE==5)  { F32("v_fmac_f32 ", ",%8,%9") }                       // VOP2, 32-bit encoding
        if constexpr (MODE==6)  { F32("v_fmaak_f32 ", ",%8,%9,0x3e11e9bf") }           // VOP2 + 32-bit literal
        if constexpr (MODE==7)  { F32("v_mul_f32 ", ",0x3fc90fdb,%8") }                // VOP2 + literal
        if constexpr (MODE==8)  { F32("v_xor_b32 ", ",%8,%9") }
        if constexpr (MODE==9)  { F32("v_mul_lo_u32 ", ",%8,%9") }
        if constexpr (MODE==10) { F32("v_mul_hi_u32 ", ",%8,%9") }
        if constexpr (MODE==11) { U64("v_mad_u64_u32 ", ",vcc,%8,%9,0") }
        if constexpr (MODE==12) { F32("v_sqrt_f32 ", ",%8") }
        if constexpr (MODE==13) { F32("v_cvt_f32_u32 ", ",%8") }
        if constexpr (MODE==14) { F32("v_cndmask_b32 ", ",%8,%9,vcc") }
        if constexpr (MODE==15) { F32("v_fma_f32 ", ",%8,%9,1.0") }                    // VOP3 with an inline constant
        if constexpr (MODE==16) { F32("v_mul_u32_u24 ", ",%8,%9") }
        if constexpr (MODE==17) { F32("v_log_f32 ", ",%8") }
        if constexpr (MODE==18) { F32("v_sin_f32 ", ",%8") }
        if constexpr (MODE==19) { F32("v_bfe_u32 ", ",%8,8,22") }
        if constexpr (MODE==20) { F32("v_alignbit_b32 ", ",%8,%9,13") }
        if constexpr (MODE==21) { F32("v_and_or_b32 ", ",%8,%9,1.0") }
        if constexpr (MODE==22) { F32("v_add_u32 ", ",%8,%9") }
        if constexpr (MODE==23) { F32("v_xad_u32 ", ",%8,%9,%9") }
        if constexpr (MODE==24) { F32("v_add3_u32 ", ",%8,%9,%9") }
        if constexpr (MODE==25) { F32("v_rcp_f32 ", ",%8") }
        if constexpr (MODE==26) { F32("v_mov_b32 ", ",%8") }
        if constexpr (MODE==27) { U64("v_lshl_add_u64 ", ",%0,0,%0 ; ") }
        if constexpr (MODE==28) { F32("v_cndmask_b32 ", ",%8,%9,s[10:11]") }          // VOP3, mask in an SGPR pair
        if constexpr (MODE==29) { F32("v_addc_co_u32 ", ",vcc,%8,%9,vcc") }
        if constexpr (MODE==30) { F32("v_cmp_lt_f32 vcc,%8,", " ; ") }                 // VOPC: writes vcc
        if constexpr (MODE==31) { F32("v_cmp_lt_f32 s[10:11],%8,", " ; ") }            // VOP3 compare into an SGPR pair
        if constexpr (MODE==32) { F32("v_bfi_b32 ", ",%8,%9,%9") }
        if constexpr (MODE==33) { F32("v_max_f32 ", ",%8,%9") }
        if constexpr (MODE==34) { F32("v_med3_f32 ", ",%8,%9,%9") }
        if constexpr (MODE==35) { REP16(asm volatile("v_cmp_lt_f32 vcc,%8,%0\n s_nop 1\n v_cndmask_b32 %0,%8,%9,vcc\n v_cmp_lt_f32 vcc,%8,%1\n s_nop 1\n v_cndmask_b32 %1,%8,%9,vcc\n"
                                                     "v_cmp_lt_f32 vcc,%8,%2\n s_nop 1\n v_cndmask_b32 %2,%8,%9,vcc\n v_cmp_lt_f32 vcc,%8,%3\n s_nop 1\n v_cndmask_b32 %3,%8,%9,vcc"
                                                     : "+v"(a0.x),"+v"(a1.x),"+v"(a2.x),"+v"(a3.x),"+v"(a4.x),"+v"(a5.x),"+v"(a6.x),"+v"(a7.x) : "v"(b.x), "v"(b.y) : "vcc");) }   // 4 x (cmp, nop, cndmask) = 8 VALU
        if constexpr (MODE==36) { F32("v_frexp_mant_f32 ", ",%8") }
        if constexpr (MODE==37) { F32("v_lshrrev_b32 ", ",23,%8") }
        if constexpr (MODE==38) { F32("v_cvt_f32_i32 ", ",%8") }
        if constexpr (MODE==39) { F32("v_mul_f32 ", ",%8,%9") }
    }
    out[blockIdx.x*blockDim.x+threadIdx.x]=a0.x+a1.x+a2.x+a3.x+a4.x+a5.x+a6.x+a7.x+a0.y+a1.y+a7.y+(float)(q0+q1+q2+q3+q4+q5+q6+q7);
}
template<int MODE> void run(const char* name){
    float* out; hipMalloc(&out, 256*8*1024*4);
    printf("%-28s", name);
    for (int wg_per_cu : {1, 2, 4}) {
        const int threads = 256, iters = 1000, grid = 256 * wg_per_cu;
        hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<grid,threads>>>(out,10); hipDeviceSynchronize();
        hipEventRecord(e0); k<MODE><<<grid,threads>>>(out,iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms,e0,e1);
        const double ns_per = ms * 1e6 / ((double)iters * 128 * wg_per_cu * threads / 64 / 4);
        printf("  waves/simd=%d : %5.2f cyc", wg_per_cu, ns_per * 2.4);      // per wave-instruction per SIMD at 2.4 GHz
    }
    printf("\n");
    hipFree(out);
}
int main(){
    run<0>("v_pk_add_f32"); run<4>("v_pk_add_f32 op_sel neg"); run<1>("v_add_f32 (VOP2)"); run<2>("v_pk_fma_f32"); run<3>("v_fma_f32 (VOP3)");
    run<15>("v_fma_f32 inline const"); run<5>("v_fmac_f32 (VOP2)"); run<6>("v_fmaak_f32 (VOP2+literal)"); run<7>("v_mul_f32 literal (VOP2)");
    run<8>("v_xor_b32"); run<22>("v_add_u32"); run<26>("v_mov_b32"); run<14>("v_cndmask_b32 vcc"); run<19>("v_bfe_u32"); run<20>("v_alignbit_b32"); run<21>("v_and_or_b32");
    run<23>("v_xad_u32"); run<24>("v_add3_u32"); run<27>("v_lshl_add_u64");
    run<16>("v_mul_u32_u24"); run<9>("v_mul_lo_u32"); run<10>("v_mul_hi_u32"); run<11>("v_mad_u64_u32");
    run<28>("v_cndmask_b32 sgpr pair"); run<29>("v_addc_co_u32"); run<30>("v_cmp_lt_f32 -> vcc"); run<31>("v_cmp_lt_f32 -> sgpr pair"); run<32>("v_bfi_b32");
    run<33>("v_max_f32"); run<34>("v_med3_f32"); run<35>("cmp + nop + cndmask (x0.5)"); run<36>("v_frexp_mant_f32"); run<37>("v_lshrrev_b32"); run<38>("v_cvt_f32_i32"); run<39>("v_mul_f32");
    run<13>("v_cvt_f32_u32"); run<12>("v_sqrt_f32"); run<25>("v_rcp_f32"); run<17>("v_log_f32"); run<18>("v_sin_f32");
    return 0;
}
