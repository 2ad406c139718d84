// throwaway: statistics of the HC parse (block.go:319-481 restated) for the design of a lock-step miss front
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
  int G=argc>2?atoi(argv[2]):8, CMAX=argc>3?atoi(argv[3]):16, K=argc>4?atoi(argv[4]):1, ADAPT=argc>5?atoi(argv[5]):0;
  uint32_t depth=131072;
  long searches=0, misses=0, hits=0, nv_hist[6]={0}, miss_simple=0;
  long front_calls=0, front_commit=0, front_zero=0, full_searches=0, hist_m[17]={0};
  long cut_hit=0, cut_complex=0, cut_dup=0, cut_end=0;
  // record per search: miss(1)/hit, simple (nv<=CMAX), hash
  for(size_t b0=0;b0<N;b0+=1<<20){
    const uint8_t*src=buf+b0; long n=(N-b0)<(1<<20)?(N-b0):(1<<20);
    memset(ht,0,sizeof ht); memset(ct,0,sizeof ct);
    long si=0, anchor=0, sn=n-14;
    // per-search log for this block
    int cap=1<<21; uint8_t*lg_miss=malloc(cap),*lg_simple=malloc(cap); uint32_t*lg_h=malloc(cap*4); int ns=0;
    while(si<sn){
      uint32_t match=ld32(src+si), h=hh(match); long m_len=0; long next=ht[h]; int nv=0;
      for(uint32_t t=depth;t>0&&next>0&&si-next<WIN;next=ct[next&0xffff],t--){
        nv++;
        if(src[next+m_len]!=src[si+m_len])continue;
        long ml=0; while(ml<sn-si){uint64_t x=ld64(src+next+ml)^ld64(src+si+ml); if(x==0)ml+=8; else{ml+=__builtin_ctzll(x)>>3;break;}}
        if(ml<4||ml<=m_len)continue; m_len=ml;
      }
      ct[si&0xffff]=ht[h]; ht[h]=(int32_t)si;
      searches++; lg_h[ns]=h; lg_miss[ns]=m_len==0; lg_simple[ns]=nv<=CMAX; ns++;
      nv_hist[nv==0?0:nv<=8?1:nv<=16?2:nv<=24?3:nv<=64?4:5]++;
      if(m_len==0){misses++; if(nv<=CMAX)miss_simple++; si+=1+((si-anchor)>>7); continue;}
      hits++;
      long p=si+1, ws=si+m_len-WIN; if(ws>p)p=ws;
      for(long e=si+m_len;p<e;p++){match=(match>>8)|((uint32_t)src[p+3]<<24); uint32_t x=hh(match); ct[p&0xffff]=ht[x]; ht[x]=(int32_t)p;}
      si+=m_len; anchor=si;
    }
    // simulate: front runs when the previous search missed; G searches; commits leading simple misses without dup hash
    int i=0; int prev_miss=0; int streak=0, gate=K; int pen=0, skip=0;
    while(i<ns){
      if(streak>=gate && skip==0){
        front_calls++; int m=0;
        for(;m<G&&i+m<ns;m++){
          if(!lg_miss[i+m]){cut_hit++;break;} if(!lg_simple[i+m]){cut_complex++;break;}
          int dup=0; for(int k=0;k<m;k++) if(lg_h[i+k]==lg_h[i+m]) dup=1; if(dup){cut_dup++;break;}
        }
        if(m==G||i+m>=ns)cut_end++;
        front_commit+=m; hist_m[m]++; if(m==0)front_zero++;
        if(ADAPT>=100){ int PM=ADAPT-100; if(m==0){pen=2*pen+1; if(pen>PM)pen=PM; skip=pen;} else if(m==1){skip=pen;} else {pen/=2; skip=0;} }
        else if(ADAPT){ if(m==0) gate=gate<ADAPT?gate+1:ADAPT; else if(m>=2) gate=K; }
        streak+=m; i+=m; if(m==G){prev_miss=1;continue;}
        if(i>=ns)break;
      }
      if(skip>0)skip--; full_searches++; prev_miss=lg_miss[i]; streak=prev_miss?streak+1:0; i++;
    }
    free(lg_miss);free(lg_simple);free(lg_h);
  }
  printf("searches %ld (%.3f/B) miss %.3f  nv: 0:%.3f 1-8:%.3f 9-16:%.3f 17-24:%.3f 25-64:%.3f >64:%.3f  simple misses %.3f of misses\n",searches,(double)searches/N,(double)misses/searches,
    (double)nv_hist[0]/searches,(double)nv_hist[1]/searches,(double)nv_hist[2]/searches,(double)nv_hist[3]/searches,(double)nv_hist[4]/searches,(double)nv_hist[5]/searches,(double)miss_simple/misses);
  printf("G=%d CMAX=%d: front calls %ld (%.3f per search), committed %ld (%.3f of searches, %.2f per call), zero-commit calls %.3f, full searches %ld (%.3f)\n",G,CMAX,front_calls,(double)front_calls/searches,front_commit,(double)front_commit/searches,(double)front_commit/front_calls,(double)front_zero/front_calls,full_searches,(double)full_searches/searches);
  printf("cost (full + 1.5 front) per search: %.3f\n",((double)full_searches+1.5*front_calls)/searches);
  printf("cuts: hit %ld complex %ld dup %ld full/end %ld; m hist:",cut_hit,cut_complex,cut_dup,cut_end); for(int m=0;m<=G;m++)printf(" %ld",hist_m[m]); printf("\n");
}
