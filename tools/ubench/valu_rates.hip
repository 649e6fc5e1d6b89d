// Issue-rate microbenchmark for the VALU instruction classes the render kernel is made of (dev tool).
// Each kernel runs `iters` x 16 independent instructions of one class per wave, WAVES waves per SIMD on every CU;
// cycles per wave-instruction per SIMD = elapsed * clock / (iters * 16 * waves_per_simd).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__ ( 256 ) void bench ( float* out, int iters, float seed, unsigned long long* cyc ) {
    unsigned long long t0 = __builtin_readcyclecounter();
    extern __shared__ float lds_dummy[];
    if ( seed == -1.f ) out[1] = lds_dummy[threadIdx.x];
    float a[16]; double d[16]; unsigned long long q[16]; unsigned u[16];
    for ( int i = 0; i < 16; ++i ) { a[i] = seed + i + threadIdx.x; d[i] = a[i]; q[i] = ( unsigned long long ) ( a[i] * 77.f ) | 1ull; u[i] = ( unsigned ) q[i]; }
    float b = seed * 0.5f + 1.f; double db = b; unsigned long long qb = 6364136223846793005ull; unsigned ub = 0x9E3779B9u + ( unsigned ) seed;
    for ( int it = 0; it < iters; ++it ) {
#define F_ADD(i)  asm volatile ( "v_add_f32 %0, %0, %1" : "+v"( a[i] ) : "v"( b ) );
#define F_MUL(i)  asm volatile ( "v_mul_f32 %0, %0, %1" : "+v"( a[i] ) : "v"( b ) );
#define F_FMA(i)  asm volatile ( "v_fma_f32 %0, %0, %1, %1" : "+v"( a[i] ) : "v"( b ) );
#define F_MIN3(i) asm volatile ( "v_min3_f32 %0, %0, %1, %1" : "+v"( a[i] ) : "v"( b ) );
#define F_RCP(i)  asm volatile ( "v_rcp_f32 %0, %0" : "+v"( a[i] ) );
#define F_SQRT(i) asm volatile ( "v_sqrt_f32 %0, %0" : "+v"( a[i] ) );
#define F_RSQ(i)  asm volatile ( "v_rsq_f32 %0, %0" : "+v"( a[i] ) );
#define D_ADD(i)  asm volatile ( "v_add_f64 %0, %0, %1" : "+v"( d[i] ) : "v"( db ) );
#define D_MUL(i)  asm volatile ( "v_mul_f64 %0, %0, %1" : "+v"( d[i] ) : "v"( db ) );
#define D_FMA(i)  asm volatile ( "v_fma_f64 %0, %0, %1, %1" : "+v"( d[i] ) : "v"( db ) );
#define D_RCP(i)  asm volatile ( "v_rcp_f64 %0, %0" : "+v"( d[i] ) );
#define D_CVT(i)  asm volatile ( "v_cvt_f64_f32 %0, %1" : "=v"( d[i] ) : "v"( a[i] ) );
#define D_CVTB(i) asm volatile ( "v_cvt_f32_f64 %0, %1" : "=v"( a[i] ) : "v"( d[i] ) );
#define I_MULLO(i) asm volatile ( "v_mul_lo_u32 %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_MULHI(i) asm volatile ( "v_mul_hi_u32 %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_MAD64(i) asm volatile ( "v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"( q[i] ) : "v"( u[i] ), "v"( ub ) : "vcc" );
#define I_ADD(i)  asm volatile ( "v_add_u32 %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_LSHLADD(i) asm volatile ( "v_lshl_add_u32 %0, %0, 3, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_CNDMASK(i) asm volatile ( "v_cndmask_b32 %0, %0, %1, vcc" : "+v"( u[i] ) : "v"( ub ) : "vcc" );
#define I_CMP(i)  asm volatile ( "v_cmp_gt_f32 vcc, %0, %1" : : "v"( a[i] ), "v"( b ) : "vcc" );
#define I_MUL24(i) asm volatile ( "v_mul_u32_u24 %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_ALIGNBIT(i) asm volatile ( "v_alignbit_b32 %0, %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_LSHR64(i) asm volatile ( "v_lshrrev_b64 %0, 18, %0" : "+v"( q[i] ) );
#define F_DIVSCALE(i) asm volatile ( "v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"( a[i] ) : "v"( b ) : "vcc" );
#define F_DIVFIXUP(i) asm volatile ( "v_div_fixup_f32 %0, %0, %1, %1" : "+v"( a[i] ) : "v"( b ) );
#define F_DIVFMAS(i) asm volatile ( "v_div_fmas_f32 %0, %0, %1, %1" : "+v"( a[i] ) : "v"( b ) : "vcc" );
#define P_MUL(i) asm volatile ( "v_pk_mul_f32 %0, %0, %1" : "+v"( d[i] ) : "v"( db ) );
#define P_ADD(i) asm volatile ( "v_pk_add_f32 %0, %0, %1" : "+v"( d[i] ) : "v"( db ) );
#define P_FMA(i) asm volatile ( "v_pk_fma_f32 %0, %0, %1, %1" : "+v"( d[i] ) : "v"( db ) );
#define V_MOV(i) asm volatile ( "v_mov_b32 %0, %1" : "=v"( u[i] ) : "v"( ub ) );
#define P_MOV(i) asm volatile ( "v_pk_mov_b32 %0, %0, %1" : "+v"( d[i] ) : "v"( db ) );
#define F_SUB(i) asm volatile ( "v_sub_f32 %0, %0, %1" : "+v"( a[i] ) : "v"( b ) );
#define F_MAX(i) asm volatile ( "v_max_f32 %0, %0, %1" : "+v"( a[i] ) : "v"( b ) );
#define C_VCC(i) asm volatile ( "v_cndmask_b32 %0, %0, %1, vcc" : "+v"( u[i] ) : "v"( ub ) );
#define C_SGPR(i) asm volatile ( "v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"( u[i] ) : "v"( ub ) );
#define F_MIN(i) asm volatile ( "v_min_f32 %0, %0, %1" : "+v"( a[i] ) : "v"( b ) );
#define I_AND(i) asm volatile ( "v_and_b32 %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_XOR(i) asm volatile ( "v_xor_b32 %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_LSHL(i) asm volatile ( "v_lshlrev_b32 %0, 3, %0" : "+v"( u[i] ) );
#define I_LSHR(i) asm volatile ( "v_lshrrev_b32 %0, 3, %0" : "+v"( u[i] ) );
#define I_ADD3(i) asm volatile ( "v_add3_u32 %0, %0, %1, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_SUB(i) asm volatile ( "v_sub_u32 %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
#define F_MED3(i) asm volatile ( "v_med3_f32 %0, %0, %1, %1" : "+v"( a[i] ) : "v"( b ) );
#define F_CVTU(i) asm volatile ( "v_cvt_f32_u32 %0, %1" : "=v"( a[i] ) : "v"( u[i] ) );
#define F_FMAC(i) asm volatile ( "v_fmac_f32 %0, %1, %1" : "+v"( a[i] ) : "v"( b ) );
#define F_MULK(i) asm volatile ( "v_mul_f32 %0, 0x3a83126f, %0" : "+v"( a[i] ) );
#define I_CMPS(i) asm volatile ( "v_cmp_gt_f32_e64 s[22:23], %0, %1" : : "v"( a[i] ), "v"( b ) : "s22", "s23" );
#define I_BFE(i) asm volatile ( "v_bfe_u32 %0, %0, 3, 5" : "+v"( u[i] ) );
#define I_ADDC(i) asm volatile ( "v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"( u[i] ) : "v"( ub ) : "vcc" );
#define C_VCC64(i) asm volatile ( "v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"( u[i] ) : "v"( ub ) );
#define CC_VCC(i) asm volatile ( "v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"( u[i] ) : "v"( a[i] ), "v"( b ), "v"( ub ) : "vcc" );
#define CC_SGPR(i) asm volatile ( "v_cmp_gt_f32_e64 s[22:23], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[22:23]" : "+v"( u[i] ) : "v"( a[i] ), "v"( b ), "v"( ub ) : "s22", "s23" );
#define CC_VCC_FAR(i) asm volatile ( "v_cmp_gt_f32 vcc, %1, %2\n\tv_add_f32 %1, %1, %2\n\tv_mul_f32 %4, %4, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"( u[i] ), "+v"( a[i] ) : "v"( b ), "v"( ub ), "v"( a[( i + 8 ) & 15] ) : "vcc" );
#define CC_SGPR_FAR(i) asm volatile ( "v_cmp_gt_f32_e64 s[22:23], %1, %2\n\tv_add_f32 %1, %1, %2\n\tv_mul_f32 %4, %4, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[22:23]" : "+v"( u[i] ), "+v"( a[i] ) : "v"( b ), "v"( ub ), "v"( a[( i + 8 ) & 15] ) : "s22", "s23" );
#define S_SALU(i) asm volatile ( "s_add_u32 s20, s20, 1" : : : "s20" );
#define F_MIX(i) asm volatile ( "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"( a[i] ) : "v"( u[i] ), "v"( b ) );
#define V_PERM(i) asm volatile ( "v_perm_b32 %0, %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_MINU(i) asm volatile ( "v_min_u32 %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
#define I_MAXU(i) asm volatile ( "v_max_u32 %0, %0, %1" : "+v"( u[i] ) : "v"( ub ) );
        // mixed streams, 48 instructions per iteration (cycles per instruction are then printed for 16 x 3 = 48): the DYNAMIC class mix of the headline kernel
        // (SQ_INSTS_VALU_* counters, profiles/r04_pmc.json: add 12 %, mul 15 %, fma 20 %, transcendental 2 %, integer 17 %, the rest compares / selects / min-max / moves)
        // and the instruction mix of the fast tree's 4-wide node step (v_fma_mix_f32, v_perm_b32, min / max, compares, selects, the sorting network's integer min / max)
        if ( OP == 100 ) { for ( int rep = 0; rep < 1; ++rep ) { F_ADD ( 0 ) F_MUL ( 1 ) F_FMA ( 2 ) I_CMPS ( 3 ) F_FMA ( 4 ) I_ADD ( 5 ) C_SGPR ( 6 ) F_MUL ( 7 ) F_FMA ( 8 ) V_MOV ( 9 ) F_ADD ( 10 ) I_AND ( 11 ) F_FMA ( 12 ) F_MIN3 ( 13 ) F_MUL ( 14 ) I_CMPS ( 15 ) F_FMA ( 0 ) I_LSHLADD ( 1 ) C_SGPR ( 2 ) F_ADD ( 3 ) F_MUL ( 4 ) F_FMA ( 5 ) I_ADD ( 6 ) I_CMPS ( 7 ) F_RCP ( 8 ) F_FMA ( 9 ) V_MOV ( 10 ) F_MUL ( 11 ) I_ADD ( 12 ) F_ADD ( 13 ) C_SGPR ( 14 ) F_FMA ( 15 ) F_MAX ( 0 ) I_CMPS ( 1 ) F_MUL ( 2 ) I_AND ( 3 ) F_FMA ( 4 ) I_LSHR ( 5 ) F_ADD ( 6 ) I_LSHLADD ( 7 ) C_SGPR ( 8 ) F_MUL ( 9 ) F_FMA ( 10 ) V_MOV ( 11 ) I_CMPS ( 12 ) F_ADD ( 13 ) I_ADD ( 14 ) F_MIN3 ( 15 ) } }
        if ( OP == 101 ) { for ( int rep = 0; rep < 1; ++rep ) { F_MIX ( 0 ) F_MIX ( 1 ) V_PERM ( 2 ) F_MIX ( 3 ) I_CMPS ( 4 ) F_MIX ( 5 ) V_PERM ( 6 ) C_SGPR ( 7 ) F_MIX ( 8 ) F_MIN3 ( 9 ) F_MIX ( 10 ) I_ADD ( 11 ) V_PERM ( 12 ) F_MIX ( 13 ) F_MAX ( 14 ) I_CMPS ( 15 ) F_MIX ( 0 ) C_SGPR ( 1 ) I_MINU ( 2 ) F_MIX ( 3 ) V_PERM ( 4 ) F_MIX ( 5 ) F_MIN3 ( 6 ) I_CMPS ( 7 ) V_MOV ( 8 ) F_MIX ( 9 ) I_MAXU ( 10 ) C_SGPR ( 11 ) V_PERM ( 12 ) I_ADD ( 13 ) F_MAX ( 14 ) I_CMPS ( 15 ) F_MIN3 ( 0 ) C_SGPR ( 1 ) I_MINU ( 2 ) V_MOV ( 3 ) I_LSHL ( 4 ) V_PERM ( 5 ) I_CMPS ( 6 ) C_SGPR ( 7 ) I_ADD ( 8 ) F_MIN3 ( 9 ) I_MAXU ( 10 ) I_AND ( 11 ) V_MOV ( 12 ) I_ADD ( 13 ) I_LSHL ( 14 ) F_MIX ( 15 ) } }
        if ( OP == 0 ) { REP16 ( F_ADD ) }
        if ( OP == 1 ) { REP16 ( F_MUL ) }
        if ( OP == 2 ) { REP16 ( F_FMA ) }
        if ( OP == 3 ) { REP16 ( F_MIN3 ) }
        if ( OP == 4 ) { REP16 ( F_RCP ) }
        if ( OP == 5 ) { REP16 ( F_SQRT ) }
        if ( OP == 6 ) { REP16 ( F_RSQ ) }
        if ( OP == 7 ) { REP16 ( D_ADD ) }
        if ( OP == 8 ) { REP16 ( D_MUL ) }
        if ( OP == 9 ) { REP16 ( D_FMA ) }
        if ( OP == 10 ) { REP16 ( D_RCP ) }
        if ( OP == 11 ) { REP16 ( D_CVT ) }
        if ( OP == 12 ) { REP16 ( D_CVTB ) }
        if ( OP == 13 ) { REP16 ( I_MULLO ) }
        if ( OP == 14 ) { REP16 ( I_MULHI ) }
        if ( OP == 15 ) { REP16 ( I_MAD64 ) }
        if ( OP == 16 ) { REP16 ( I_ADD ) }
        if ( OP == 17 ) { REP16 ( I_LSHLADD ) }
        if ( OP == 18 ) { REP16 ( I_CNDMASK ) }
        if ( OP == 19 ) { REP16 ( I_CMP ) }
        if ( OP == 20 ) { REP16 ( I_MUL24 ) }
        if ( OP == 21 ) { REP16 ( I_ALIGNBIT ) }
        if ( OP == 22 ) { REP16 ( I_LSHR64 ) }
        if ( OP == 23 ) { REP16 ( F_DIVSCALE ) }
        if ( OP == 24 ) { REP16 ( F_DIVFIXUP ) }
        if ( OP == 25 ) { REP16 ( F_DIVFMAS ) }
        if ( OP == 26 ) { REP16 ( S_SALU ) }
        if ( OP == 27 ) { REP16 ( P_MUL ) }
        if ( OP == 28 ) { REP16 ( P_ADD ) }
        if ( OP == 29 ) { REP16 ( P_FMA ) }
        if ( OP == 30 ) { REP16 ( V_MOV ) }
        if ( OP == 31 ) { REP16 ( P_MOV ) }
        if ( OP == 32 ) { REP16 ( F_SUB ) }
        if ( OP == 33 ) { REP16 ( F_MAX ) }
        if ( OP == 34 ) { REP16 ( C_VCC ) }
        if ( OP == 35 ) { REP16 ( C_SGPR ) }
        if ( OP == 36 ) { REP16 ( F_MIN ) }
        if ( OP == 37 ) { REP16 ( I_AND ) }
        if ( OP == 38 ) { REP16 ( I_XOR ) }
        if ( OP == 39 ) { REP16 ( I_LSHL ) }
        if ( OP == 40 ) { REP16 ( I_LSHR ) }
        if ( OP == 41 ) { REP16 ( I_ADD3 ) }
        if ( OP == 42 ) { REP16 ( I_SUB ) }
        if ( OP == 43 ) { REP16 ( F_MED3 ) }
        if ( OP == 44 ) { REP16 ( F_CVTU ) }
        if ( OP == 45 ) { REP16 ( F_FMAC ) }
        if ( OP == 46 ) { REP16 ( F_MULK ) }
        if ( OP == 47 ) { REP16 ( I_CMPS ) }
        if ( OP == 48 ) { REP16 ( I_BFE ) }
        if ( OP == 49 ) { REP16 ( I_ADDC ) }
        if ( OP == 50 ) { REP16 ( C_VCC64 ) }
        if ( OP == 51 ) { REP16 ( CC_VCC ) }
        if ( OP == 52 ) { REP16 ( CC_SGPR ) }
        if ( OP == 53 ) { REP16 ( CC_VCC_FAR ) }
        if ( OP == 54 ) { REP16 ( CC_SGPR_FAR ) }
    }
    float s = 0; for ( int i = 0; i < 16; ++i ) s += a[i] + ( float ) d[i] + ( float ) q[i] + ( float ) u[i];
    if ( s == 12345.678f ) out[0] = s;
    if ( blockIdx.x == 0 && threadIdx.x == 0 ) cyc[0] = __builtin_readcyclecounter() - t0;
}

template <int OP> void run ( const char* name, int waves_per_simd, float* out, double clock_ghz, int cus ) {
    const int per_iter = OP >= 100 ? 48 : 16;
    const int iters = OP >= 100 ? 20000 : 60000;
    const int rounds = 6;
    dim3 grid ( cus * waves_per_simd * rounds ), block ( 256 );
    size_t lds = waves_per_simd == 5 ? 31000 : ( 160 * 1024 ) / waves_per_simd - 512;      // exactly waves_per_simd blocks fit a CU (5: the CU does not hand out all of its 160 KB)
    hipEvent_t e0, e1; hipEventCreate ( &e0 ); hipEventCreate ( &e1 );
    static unsigned long long* dc = nullptr; if ( !dc ) hipMalloc ( &dc, 8 );
    hipFuncSetAttribute ( ( const void* ) bench<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 );
    bench<OP><<<grid, block, lds>>> ( out, 10, 1.f, dc );
    hipDeviceSynchronize();
    hipEventRecord ( e0 );
    bench<OP><<<grid, block, lds>>> ( out, iters, 1.f, dc );
    hipEventRecord ( e1 ); hipEventSynchronize ( e1 );
    float ms; hipEventElapsedTime ( &ms, e0, e1 );
    unsigned long long hc = 0; hipMemcpy ( &hc, dc, 8, hipMemcpyDeviceToHost );
    double cyc = ( double ) hc / ( ( double ) iters * per_iter * waves_per_simd );
    double ginst = ( double ) iters * per_iter * waves_per_simd * rounds / ( ms * 1e-3 ) / 1e9;
    printf ( "%-14s waves/SIMD %d  %.3f ms  %.3f G wave-instr/s per SIMD = %.2f cycles at 2.4 GHz (one block: %.2f counter ticks per instr)\n", name, waves_per_simd, ms, ginst, 2.4 / ginst, cyc );
}

int main ( int argc, char** argv ) {
    hipDeviceProp_t p; hipGetDeviceProperties ( &p, 0 );
    double ghz = p.clockRate * 1e-6;
    printf ( "%s CUs %d clock %.2f GHz\n", p.name, p.multiProcessorCount, ghz );
    float* out; hipMalloc ( &out, 4 );
#define R(op, n) run<op> ( n, w, out, ghz, p.multiProcessorCount );
    if ( argc > 1 && std::string ( argv[1] ) == "select" ) {       // the select forms: mask in VCC against mask in an SGPR pair (the 51 - 54 streams hold 2 / 4 instructions per slot: divide their cycles accordingly)
        for ( int w : { 4 } ) { R ( 34, "v_cndmask vcc" ) R ( 50, "v_cndmask_e64 vcc" ) R ( 35, "v_cndmask sgpr" ) R ( 51, "cmp+cndmask vcc (x2)" ) R ( 52, "cmp+cndmask sgpr (x2)" ) R ( 53, "cmp,add,mul,cndmask vcc (x4)" ) R ( 54, "cmp,add,mul,cndmask sgpr (x4)" ) }
        return 0;
    }
    if ( argc > 1 && std::string ( argv[1] ) == "mix" ) {          // only the mixed streams, at the occupancies the kernels run at
        for ( int w : { 4, 5 } ) { R ( 100, "mix: headline kernel" ) R ( 101, "mix: 4-wide node step" ) R ( 2, "v_fma_f32" ) R ( 0, "v_add_f32" ) R ( 35, "v_cndmask sgpr" ) }
        return 0;
    }
    for ( int w : { 4 } ) {
        R ( 0, "v_add_f32" ) R ( 1, "v_mul_f32" ) R ( 2, "v_fma_f32" ) R ( 3, "v_min3_f32" ) R ( 4, "v_rcp_f32" ) R ( 5, "v_sqrt_f32" ) R ( 6, "v_rsq_f32" )
        R ( 7, "v_add_f64" ) R ( 8, "v_mul_f64" ) R ( 9, "v_fma_f64" ) R ( 10, "v_rcp_f64" ) R ( 11, "v_cvt_f64_f32" ) R ( 12, "v_cvt_f32_f64" )
        R ( 13, "v_mul_lo_u32" ) R ( 14, "v_mul_hi_u32" ) R ( 15, "v_mad_u64_u32" ) R ( 16, "v_add_u32" ) R ( 17, "v_lshl_add_u32" )  R ( 19, "v_cmp_f32" )
        R ( 20, "v_mul_u32_u24" ) R ( 21, "v_alignbit" ) R ( 22, "v_lshrrev_b64" ) R ( 23, "v_div_scale" ) R ( 24, "v_div_fixup" ) R ( 25, "v_div_fmas" ) R ( 27, "v_pk_mul_f32" ) R ( 28, "v_pk_add_f32" ) R ( 29, "v_pk_fma_f32" ) R ( 30, "v_mov_b32" ) R ( 31, "v_pk_mov_b32" ) R ( 32, "v_sub_f32" ) R ( 33, "v_max_f32" ) R ( 34, "v_cndmask vcc" ) R ( 35, "v_cndmask sgpr" ) R ( 36, "v_min_f32" ) R ( 37, "v_and_b32" ) R ( 38, "v_xor_b32" ) R ( 39, "v_lshlrev_b32" ) R ( 40, "v_lshrrev_b32" ) R ( 41, "v_add3_u32" ) R ( 42, "v_sub_u32" ) R ( 43, "v_med3_f32" ) R ( 44, "v_cvt_f32_u32" ) R ( 45, "v_fmac_f32" ) R ( 46, "v_mul_f32 lit" ) R ( 47, "v_cmp e64 sgpr" ) R ( 48, "v_bfe_u32" ) R ( 49, "v_addc_co_u32" )
    }
    return 0;
}
