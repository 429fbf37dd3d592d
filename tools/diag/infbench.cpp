#include "../../nextpolish_amd/csrc/np_inflate.h"
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <time.h>
static double now(){timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); std::vector<uint8_t> d(60<<20); size_t n=fread(d.data(),1,d.size(),f); fclose(f);
  struct B{size_t off,clen; uint32_t isize;}; std::vector<B> bs; size_t p=0;
  while(p+28<n && bs.size()<800){ uint16_t bsize; memcpy(&bsize,&d[p+16],2); size_t tot=bsize+1; uint32_t isz; memcpy(&isz,&d[p+tot-4],4); bs.push_back({p+18,tot-26,isz}); p+=tot; }
  std::vector<uint8_t> out(65536); size_t totb=0; for(auto&b:bs) totb+=b.isize;
  for(int rep=0;rep<2;++rep){
   double t=now(); for(int k=0;k<5;++k) for(auto&b:bs) if(!np::inflate_raw(&d[b.off],b.clen,out.data(),b.isize)) {printf("reject\n");return 1;}
   double t1=(now()-t)/5;
   t=now(); for(int k=0;k<5;++k) for(auto&b:bs){ z_stream zs; memset(&zs,0,sizeof zs); inflateInit2(&zs,-15); zs.next_in=&d[b.off]; zs.avail_in=b.clen; zs.next_out=out.data(); zs.avail_out=b.isize; inflate(&zs,Z_FINISH); inflateEnd(&zs);} 
   double t2=(now()-t)/5;
   printf("%zu blocks %.1f MB: own %.0f MB/s, zlib %.0f MB/s\n",bs.size(),totb/1e6,totb/1e6/t1,totb/1e6/t2);
  }
}
