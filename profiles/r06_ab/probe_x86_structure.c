#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <xmmintrin.h>
static inline uint32_t hw_rcp(uint32_t u){ float f; memcpy(&f,&u,4); float o=_mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(f))); uint32_t v; memcpy(&v,&o,4); return v; }
static inline uint32_t hw_rsq(uint32_t u){ float f; memcpy(&f,&u,4); float o=_mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(f))); uint32_t v; memcpy(&v,&o,4); return v; }
int main(){
  for(int which=0; which<3; which++){
    uint32_t e = which==2 ? 128u : 127u;
    uint32_t prev=0; long distinct=0; long minrun=1<<30, maxrun=0, run=0; int mintz=23; uint32_t orv=0; long nonmono=0;
    for(uint32_t m=0;m<(1u<<23);m++){
      uint32_t u=(e<<23)|m; uint32_t v= which? hw_rsq(u):hw_rcp(u);
      orv |= v;
      if(m==0||v!=prev){ if(m){ if(run<minrun)minrun=run; if(run>maxrun)maxrun=run; int tz=__builtin_ctz(m); if(tz<mintz)mintz=tz; if(v>prev) nonmono++; } distinct++; run=0; }
      run++; prev=v;
    }
    printf("which=%d e=%u distinct=%ld minrun=%ld maxrun=%ld min-boundary-tz=%d or=%08x nonmonotone=%ld\n",which,e,distinct,minrun,maxrun,mintz,orv,nonmono);
  }
  // exponent independence: rcp(2^k * x) == 2^-k rcp(x) ?
  long bad=0; for(uint32_t m=0;m<(1u<<23);m+=97){ uint32_t a=hw_rcp((127u<<23)|m), b=hw_rcp((130u<<23)|m); if(((a>>23)-(b>>23))!=3 || (a&0x7fffff)!=(b&0x7fffff)) bad++; }
  printf("rcp exponent-shift mismatches %ld\n",bad);
  bad=0; for(uint32_t m=0;m<(1u<<23);m+=97){ uint32_t a=hw_rsq((127u<<23)|m), b=hw_rsq((131u<<23)|m); if(((a>>23)-(b>>23))!=2 || (a&0x7fffff)!=(b&0x7fffff)) bad++; }
  printf("rsqrt exponent-shift(4) mismatches %ld\n",bad);
  // first few boundaries
  uint32_t prev2=hw_rcp(127u<<23); int shown=0; for(uint32_t m=1;m<(1u<<23)&&shown<12;m++){ uint32_t v=hw_rcp((127u<<23)|m); if(v!=prev2){ printf("rcp boundary m=%06x -> %08x\n",m,v); prev2=v; shown++; } }
  return 0;
}
