// oracle/check_trig.c — TEST INFRASTRUCTURE.  The RoPE kernels (llama.cpp_test_amd/csrc/pkv_rowcodec.h: libm_sinf / libm_cosf)
// restate the host libm's cosf / sinf (glibc >= 2.28: the ARM optimized-routines algorithm, double-precision internal)
// so that the device produces the reference's (cos, sin) bit for bit.  This file holds the SAME restatement in plain C and
// compares it with the cosf / sinf of the libm it is linked to: over the RoPE angles of the tested geometries (both signs)
// and a strided sweep of every finite float.  Prints the mismatch counts; exit status 0 iff all are zero.
//   gcc -O2 -ffp-contract=off -o check_trig check_trig.c -lm && ./check_trig [stride]
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#pragma STDC FP_CONTRACT OFF
typedef struct { double sign[4]; double hpi_inv, hpi, c0,c1,c2,c3,c4, s1,s2,s3; } sincos_t;
static const sincos_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static const uint32_t inv_pio4[24] = {0xa2,0xa2f9,0xa2f983,0xa2f9836e,0xf9836e4e,0x836e4e44,0x6e4e4415,0x4e441529,0x441529fc,0x1529fc27,0x29fc2757,0xfc2757d1,0x2757d1f5,0x57d1f534,0xd1f534dd,0xf534ddc0,0x34ddc0db,0xddc0db62,0xc0db6295,0xdb629599,0x6295993c,0x95993c43,0x993c4390,0x3c439041};
static const double pi63 = 0x1.921FB54442D18p-62;
static inline uint32_t asuint(float f){uint32_t u; memcpy(&u,&f,4); return u;}
static inline uint32_t abstop12(float x){ return (asuint(x)>>20)&0x7ff; }
static inline float sinf_poly(double x, double x2, const sincos_t*p, int n){
  double x3,x4,x6,x7,s,c,c1,c2,s1;
  if ((n&1)==0){ x3=x*x2; s1=p->s2+x2*p->s3; x7=x3*x2; s=x+x3*p->s1; return s+x7*s1; }
  else { x4=x2*x2; c2=p->c3+x2*p->c4; c1=p->c0+x2*p->c1; x6=x4*x2; c=c1+x4*p->c2; return c+x6*c2; } }
static inline double reduce_fast(double x, const sincos_t*p, int*np){ double r=x*p->hpi_inv; int n=((int32_t)r+0x800000)>>24; *np=n; return x-n*p->hpi; }
static inline double reduce_large(uint32_t xi, int*np){ const uint32_t*arr=&inv_pio4[(xi>>26)&15]; int shift=(xi>>23)&7; uint64_t n,res0,res1,res2;
  xi=(xi&0xffffff)|0x800000; xi<<=shift; res0=xi*arr[0]; res1=(uint64_t)xi*arr[4]; res2=(uint64_t)xi*arr[8]; res0=(res2>>32)|(res0<<32); res0+=res1;
  n=(res0+(1ULL<<61))>>62; res0-=n<<62; double x=(int64_t)res0; *np=n; return x*pi63; }
float my_sinf(float y){ double x=y,s; int n; const sincos_t*p=&T[0];
  if (abstop12(y)<abstop12(0x1.921FB6p-1f)){ s=x*x; if (abstop12(y)<abstop12(0x1p-12f)) return y; return sinf_poly(x,s,p,0);} 
  else if (abstop12(y)<abstop12(120.0f)){ x=reduce_fast(x,p,&n); s=p->sign[n&3]; if(n&2)p=&T[1]; return sinf_poly(x*s,x*x,p,n);} 
  else if (abstop12(y)<abstop12(INFINITY)){ uint32_t xi=asuint(y); int sign=xi>>31; x=reduce_large(xi,&n); s=p->sign[(n+sign)&3]; if((n+sign)&2)p=&T[1]; return sinf_poly(x*s,x*x,p,n);} 
  return y-y; }
float my_cosf(float y){ double x=y,s; int n; const sincos_t*p=&T[0];
  if (abstop12(y)<abstop12(0x1.921FB6p-1f)){ s=x*x; if (abstop12(y)<abstop12(0x1p-12f)) return 1.0f; return sinf_poly(x,s,p,1);} 
  else if (abstop12(y)<abstop12(120.0f)){ x=reduce_fast(x,p,&n); s=p->sign[n&3]; if(n&2)p=&T[1]; return sinf_poly(x*s,x*x,p,n^1);} 
  else if (abstop12(y)<abstop12(INFINITY)){ uint32_t xi=asuint(y); int sign=xi>>31; x=reduce_large(xi,&n); s=p->sign[(n+sign)&3]; if((n+sign)&2)p=&T[1]; return sinf_poly(x*s,x*x,p,n^1);} 
  return y-y; }
int main(int argc, char ** argv){ long n=0, mc=0, ms=0, bad=0; uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 53; float base=500000.0f; float ts = powf(base, -2.0f/128);
  for (int pos=0; pos<40000; pos++){ float th=(float)pos; for(int ip=0; ip<64; ip++){ n++; mc += cosf(th)!=my_cosf(th); ms += sinf(th)!=my_sinf(th); mc += cosf(-th)!=my_cosf(-th); ms += sinf(-th)!=my_sinf(-th); th*=ts; } }
  printf("rope: n=%ld cos mm=%ld sin mm=%ld\n", n, mc, ms); bad += mc + ms;
  // broad sweep over float bit patterns
  n=mc=ms=0; for (uint64_t u=0; u<0x7f800000ull; u+=stride){ float f; uint32_t v=(uint32_t)u | ((u&1)?0x80000000u:0); memcpy(&f,&v,4); n++; mc+= cosf(f)!=my_cosf(f); ms+= sinf(f)!=my_sinf(f); }
  printf("sweep: n=%ld cos mm=%ld sin mm=%ld\n", n, mc, ms); bad += mc + ms; return bad != 0; }
