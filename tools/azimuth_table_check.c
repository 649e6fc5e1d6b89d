/* Can a SMALL azimuth table plus one exact refinement step reproduce the bits of the 2^24-entry table (DevScene::sincos24, filled by the glibc-faithful tdm_sincosf_pair)?
   No: glibc's sinf / cosf are not correctly rounded, so ANY accurate reconstruction -- here the best case, a 4,096-entry double table and the angle addition formulas evaluated with
   the double-precision libm -- lands on the correctly rounded float and differs from glibc's on 1.3 % of the arguments. gcc -O2 tools/azimuth_table_check.c -lm && ./a.out
   (result: profiles/r04_measurements/azimuth_table_reduction.log) */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
int main(void){
    const float PI_F = 3.1416926535f; const float c = 2*PI_F;
    static double S[4096], C[4096];
    for (int h=0;h<4096;++h){ double y=(double)c*(double)h/4096.0; S[h]=sin(y); C[h]=cos(y);}   /* y exact: c has 24 bits, h 12 */
    long bad_s=0,bad_c=0, bad_cr_s=0, bad_cr_c=0;
    for (uint32_t k=0;k<(1u<<24);++k){
        float e = (float)k * 5.9604645e-8f;            /* u24 * 2^-24, exact */
        float x = c * e;                               /* the samplers' argument, one float rounding */
        float s = sinf(x), co = cosf(x);
        int h = k>>12; double yh=(double)c*(double)h/4096.0; double z=(double)x-yh;   /* exact */
        double sz=sin(z), cz=cos(z);
        float s2=(float)(S[h]*cz+C[h]*sz), c2=(float)(C[h]*cz-S[h]*sz);
        if (memcmp(&s,&s2,4)) ++bad_s; if (memcmp(&co,&c2,4)) ++bad_c;
        float s3=(float)sin((double)x), c3=(float)cos((double)x);    /* correctly rounded (double libm then one rounding): what any accurate short-cut converges to */
        if (memcmp(&s,&s3,4)) ++bad_cr_s; if (memcmp(&co,&c3,4)) ++bad_cr_c;
    }
    printf("2^24 arguments: table(4096)+angle-addition in double differs from glibc sinf on %ld, cosf on %ld; (float)sin((double)x) differs on %ld / %ld\n",bad_s,bad_c,bad_cr_s,bad_cr_c);
    return 0;
}
