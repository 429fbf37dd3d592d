#include "../../nextpolish_amd/csrc/np_threads.h"
#include <cstdio>
#include <time.h>
int main(){ timespec a,b; clock_gettime(CLOCK_MONOTONIC,&a); std::atomic<long> s{0};
 np::parallel_for(64,1,[&](size_t lo,size_t hi){ long x=0; for(size_t i=lo;i<hi;++i) for(long k=0;k<50000000;++k) x+=k^i; s+=x;});
 clock_gettime(CLOCK_MONOTONIC,&b); printf("%u threads %.2f s %ld\n", np::host_threads(), (b.tv_sec-a.tv_sec)+(b.tv_nsec-a.tv_nsec)*1e-9, s.load()); }
