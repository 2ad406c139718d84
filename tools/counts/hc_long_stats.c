// throwaway: long chain walks of the HC parse: how many have a window in which every position was inserted
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define WIN 65536
static uint32_t ld32(const uint8_t*p){uint32_t v;memcpy(&v,p,4);return v;}
static uint64_t ld64(const uint8_t*p){uint64_t v;memcpy(&v,p,8);return v;}
static uint32_t hh(uint32_t x){return (x*2654435761u)>>16;}
static int32_t ht[65536], ct[65536];
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); size_t N=8<<20; uint8_t*buf=malloc(N+16); N=fread(buf,1,N,f); fclose(f);
  uint32_t depth=131072;
  long searches=0,longs=0,long_clean=0; double cand_all=0,cand_long=0,cand_long_clean=0; long distinct_vals_long=0;
  long hist_val[4]={0};
  for(size_t b0=0;b0<N;b0+=1<<20){
    const uint8_t*src=buf+b0; long n=(N-b0)<(1<<20)?(N-b0):(1<<20);
    memset(ht,0,sizeof ht); memset(ct,0,sizeof ct);
    long si=0, anchor=0, sn=n-14, last_skip=-1;
    while(si<sn){
      uint32_t match=ld32(src+si), h=hh(match); long m_len=0; long next=ht[h]; long nv=0, same=0;
      for(uint32_t t=depth;t>0&&next>0&&si-next<WIN;next=ct[next&0xffff],t--){
        nv++; if(ld32(src+next)==match)same++;
        if(src[next+m_len]!=src[si+m_len])continue;
        long ml=0; while(ml<sn-si){uint64_t x=ld64(src+next+ml)^ld64(src+si+ml); if(x==0)ml+=8; else{ml+=__builtin_ctzll(x)>>3;break;}}
        if(ml<4||ml<=m_len)continue; m_len=ml;
        if(m_len>=((sn-si+7)&~7))break;
      }
      ct[si&0xffff]=ht[h]; ht[h]=(int32_t)si;
      searches++; cand_all+=nv;
      if(nv>512){longs++; cand_long+=nv; int clean=(last_skip<0)||(si-last_skip>=WIN); if(clean){long_clean++;cand_long_clean+=nv;}
        hist_val[match==0?0:(match==0xffffffffu?1:((match&0xffff)==(match>>16)?2:3))]++; if(same<nv)distinct_vals_long++; }
      if(m_len==0){long step=1+((si-anchor)>>7); if(step>1)last_skip=si; si+=step; continue;}
      long p=si+1, ws=si+m_len-WIN; if(ws>p)p=ws;
      for(long e=si+m_len;p<e;p++){match=(match>>8)|((uint32_t)src[p+3]<<24); uint32_t x=hh(match); ct[p&0xffff]=ht[x]; ht[x]=(int32_t)p;}
      si+=m_len; anchor=si;
    }
  }
  printf("searches %ld, long (>512 cand) %ld (%.4f), of them clean window %ld (%.3f); candidates: all %.3g, long %.3g (%.3f), long+clean %.3g (%.3f of long)\n",
    searches,longs,(double)longs/searches,long_clean,(double)long_clean/(longs?longs:1),cand_all,cand_long,cand_long/cand_all,cand_long_clean,cand_long_clean/(cand_long?cand_long:1));
  printf("long searches by value: zero %ld, ones %ld, 2-periodic %ld, other %ld; with hash collisions in chain %ld\n",hist_val[0],hist_val[1],hist_val[2],hist_val[3],distinct_vals_long);
}
