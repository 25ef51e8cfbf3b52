/**
 * tests/helpers/fuzz_ctl.c -- hostile input for the host control plane of
 * libh264.so (parameter sets, SEI, AUD, filler, unknown types, and -- with
 * flags = 0 -- slice headers, which the library parses on the host): pseudo-
 * random NAL units through h264_reader_parse_nalu, every NAL unit dumped to
 * JSON now and then (header-level dump: no GPU involved).  Built and run under
 * AddressSanitizer + UBSan by tests/test_host_sanitizers.py (CPU build only).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <h264/h264.h>
#include "fuzz_seeds.h"
/* fuzz the host control plane: random bytes as non-slice NAL units */
static void cb_sps(struct h264_ctx *c, const uint8_t *b, size_t l, const struct h264_sps *s, void *u){(void)c;(void)b;(void)l;(void)s;(void)u;}
int main(int argc, char **argv){
  /* the host slice-header parser without a GPU behind it (reader.c) */
  setenv("H264_AMD_TEST_HEADERS_WITHOUT_ENGINE","1",1);
  struct h264_ctx_cbs cbs; memset(&cbs,0,sizeof cbs); cbs.sps=cb_sps;
  struct h264_reader *rd; h264_reader_new(&cbs,NULL,&rd);
  unsigned s=12345; uint8_t buf[512];
  static const uint8_t types[]={6,7,8,9,12,14,20,6,6,7,8,1,5,1,1,5};
  /* a sane SPS / PPS pair to begin with (and again every 256 iterations):
   * slice headers need parameter sets to be parsed against, sane or hostile */
  static const uint8_t sps0[]={0x67,0x64,0x00,0x28,0xac,0xb4,0x03,0xc0,0x11,0x32};
  static const uint8_t pps0[]={0x68,0xee,0x3c,0x80};
  struct h264_dump *dump=NULL; struct h264_dump_cfg dcfg; memset(&dcfg,0,sizeof dcfg);
  dcfg.type=H264_DUMP_TYPE_JSON; h264_dump_new(&dcfg,&dump);
  long ok=0,bad=0;
  int iters = argc > 1 ? atoi(argv[1]) : 50000;
  for(int it=0;it<iters;it++){
    size_t n=1+(s>>8)%300; s=s*1103515245+12345;
    for(size_t i=0;i<n;i++){ s=s*1103515245+12345; buf[i]=(s>>16)&((s>>28)&1?0xff:0x07);}
    s=s*1103515245+12345;
    buf[0]=(uint8_t)(((s>>9)&3)<<5 | types[(s>>16)%sizeof types]);
    if((it&255)==0){ h264_reader_parse_nalu(rd,0,sps0,sizeof sps0); h264_reader_parse_nalu(rd,0,pps0,sizeof pps0); }
    if((it&3)==3){ /* a generated slice header with a few bits flipped, maybe cut short */
      const struct fuzz_seed_set *ss=&fuzz_seed_sets[(s>>11)%(sizeof fuzz_seed_sets/sizeof fuzz_seed_sets[0])];
      s=s*1103515245+12345;
      /* mostly under the parameter sets it was made for, now and then under
       * whatever the random NAL units left behind */
      if((it&31)!=3){ h264_reader_parse_nalu(rd,0,ss->sps,ss->sps_len); h264_reader_parse_nalu(rd,0,ss->pps,ss->pps_len); }
      size_t k=(s>>9)%ss->n; n=ss->lens[k]; memcpy(buf,ss->slices[k],n);
      s=s*1103515245+12345; unsigned flips=(s>>13)&3;
      for(unsigned f=0;f<flips;f++){ s=s*1103515245+12345; size_t bit=(s>>8)%(8*n); if(bit>=8) buf[bit>>3]^=(uint8_t)(1u<<(bit&7)); }
      s=s*1103515245+12345; if(((s>>20)&7)==0) n=2+(s>>8)%(n-2);
    }
    int r=h264_reader_parse_nalu(rd,0,buf,n);
    if(r==0) ok++; else bad++;
    if(r==0 && dump && (it&7)==0){ const char *js=NULL;
      if(h264_dump_nalu(dump,h264_reader_get_ctx(rd),0)==0) h264_dump_get_json_str(dump,&js); }
    if((it&1023)==0){ struct h264_info info; h264_ctx_get_info(h264_reader_get_ctx(rd),&info);}
  }
  if(dump) h264_dump_destroy(dump);
  h264_reader_destroy(rd);
  printf("ok %ld bad %ld\n",ok,bad); return 0; }
