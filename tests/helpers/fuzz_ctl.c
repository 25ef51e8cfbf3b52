/**
 * tests/helpers/fuzz_ctl.c -- hostile input for the host control plane of
 * libh264.so (parameter sets, SEI, AUD, filler, unknown types): pseudo-random
 * NAL units through h264_reader_parse_nalu.  Built and run under
 * AddressSanitizer + UBSan by tests/test_host_sanitizers.py (CPU build only).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <h264/h264.h>
/* fuzz the host control plane: random bytes as non-slice NAL units */
static void cb_sps(struct h264_ctx *c, const uint8_t *b, size_t l, const struct h264_sps *s, void *u){(void)c;(void)b;(void)l;(void)s;(void)u;}
int main(int argc, char **argv){
  struct h264_ctx_cbs cbs; memset(&cbs,0,sizeof cbs); cbs.sps=cb_sps;
  struct h264_reader *rd; h264_reader_new(&cbs,NULL,&rd);
  unsigned s=12345; uint8_t buf[512];
  static const uint8_t types[]={6,7,8,9,12,14,20,6,6,7,8};
  long ok=0,bad=0;
  int iters = argc > 1 ? atoi(argv[1]) : 50000;
  for(int it=0;it<iters;it++){
    size_t n=1+(s>>8)%300; s=s*1103515245+12345;
    for(size_t i=0;i<n;i++){ s=s*1103515245+12345; buf[i]=(s>>16)&((s>>28)&1?0xff:0x07);}
    s=s*1103515245+12345;
    buf[0]=(uint8_t)(((s>>9)&3)<<5 | types[(s>>16)%sizeof types]);
    int r=h264_reader_parse_nalu(rd,0,buf,n);
    if(r==0) ok++; else bad++;
    if((it&1023)==0){ struct h264_info info; h264_ctx_get_info(h264_reader_get_ctx(rd),&info);}
  }
  h264_reader_destroy(rd);
  printf("ok %ld bad %ld\n",ok,bad); return 0; }
