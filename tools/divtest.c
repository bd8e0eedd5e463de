/* Checks the reciprocal-based exact division of svo_kernels.hip (div_by_recip) against IEEE a / d.
 * build: gcc -O2 -mfma -ffp-contract=off -o divtest tools/divtest.c -lm
 * run:   ./divtest <samples> <min_exp_a> <max_exp_a> <min_exp_d> <max_exp_d>   e.g. ./divtest 300000000 -53 3 -40 40 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline uint32_t rng(uint64_t *s){ *s ^= *s<<13; *s ^= *s>>7; *s ^= *s<<17; return (uint32_t)(*s>>16); }
static inline float mk(uint32_t bits){ float f; memcpy(&f,&bits,4); return f; }
static inline float fdiv(float a, float d, float y){
  float q0 = a*y;
  float r0 = fmaf(-d,q0,a);
  float q1 = fmaf(r0,y,q0);
  float r1 = fmaf(-d,q1,a);
  float q2 = fmaf(r1,y,q1);
  return q2;
}
int main(int argc,char**argv){
  uint64_t s=0x9E3779B97F4A7C15ull; long n=atol(argv[1]); long bad=0, bad1=0;
  int elo_a=atoi(argv[2]), ehi_a=atoi(argv[3]), elo_d=atoi(argv[4]), ehi_d=atoi(argv[5]);
  for(long i=0;i<n;i++){
    uint32_t ma=rng(&s)&0x7FFFFF, md=rng(&s)&0x7FFFFF;
    if ((i&7)==0) md = (rng(&s)&1)? 0x7FFFFF : (rng(&s)&0xF);      // stress extreme mantissas
    if ((i&15)==1) ma = (rng(&s)&1)? 0x7FFFFF : (rng(&s)&0xF);
    int ea = elo_a + (int)(rng(&s)%(uint32_t)(ehi_a-elo_a+1)), ed = elo_d + (int)(rng(&s)%(uint32_t)(ehi_d-elo_d+1));
    float a=mk(((uint32_t)(ea+127)<<23)|ma|((rng(&s)&1)<<31)), d=mk(((uint32_t)(ed+127)<<23)|md|((rng(&s)&1)<<31));
    float y=1.0f/d; float want=a/d; float got=fdiv(a,d,y);
    float q0=a*y; float q1=fmaf(fmaf(-d,q0,a),y,q0);
    if (memcmp(&want,&got,4)) { if(bad<5) printf("BAD a=%a d=%a want=%a got=%a\n",a,d,want,got); bad++; }
    if (memcmp(&want,&q1,4)) bad1++;
  }
  printf("n=%ld bad=%ld (single-correction mismatches: %ld)\n",n,bad,bad1);
  return bad!=0;
}
