#!/usr/bin/env python3
"""tools/phase_timers_cabac.py on REAL content: cycles per phase of the CABAC
macroblock walk over the slices of imageio's cockatoo.mp4 (60 copies of the
clip in one launch), from the -DCS_PHASE_TIMERS build
(tools/build_variant.sh ctimers ...).  Run on the GPU box."""
import os, sys, glob, ctypes as C
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests"); sys.path.insert(0, ROOT+"/tools")
os.environ["H264_AMD_ENGINE_LIB"]=ROOT+"/libh264_amd/lib/variants/libengine_ctimers.so"
import numpy as np, libh264_amd, mp4_demux, orc, streams
from libh264_amd import Params
from libh264_amd.ctypes_abi import TIER_T0
import phase_timers_cabac as pt
path=glob.glob("/opt/conda/lib/python3*/site-packages/imageio/resources/images/cockatoo.mp4")[0]
sps,pps,samples=mp4_demux.video_nals(path)
head=[bytes(x) for x in sps]+[bytes(x) for x in pps]
nals=[bytes(n) for s in samples for n in s if (n[0]&31) in (1,5)]
q=C.c_int.in_dll(orc.lib(),"orc_quirk_x264_444_cbf"); q.value=1
ref=orc.parse_nalus(head+nals, flags=0x101, want=0); q.value=0
params=Params(); params.put_sps(ref.sps[0]); epoch=params.put_pps(ref.pps[0])
copies=60
buf,offs,sizes=streams.pack_nals(nals*copies)
n=len(offs)
descs=np.zeros(n,dtype=[("offset","<u8"),("size","<u4"),("epoch","<u4")])
descs["offset"],descs["size"],descs["epoch"]=offs,sizes,epoch
eng=libh264_amd.Engine(0)
d_buf,d_descs=eng.upload(buf),eng.upload(descs)
out=None
for _ in range(2):
    out=eng.decode_slices(params,d_buf,descs,reader_flags=0x301,tier=TIER_T0,mb_capacity=3600,want_headers=True,d_descs=d_descs,outputs=out)
ms=eng.last_kernel_ms("cabac")
NP=len(pt.PHASES); acc=np.zeros(32); cnt=np.zeros(32); take=0
for i in range(0,n,max(1,n//256)):
    w=eng.fetch_mb(out,i,64).reshape(-1).view(np.uint32)[:64]; acc+=w[:32]; cnt+=w[32:64]; take+=1
mbs=3600.0*take; mark=acc[NP-1]/max(cnt[NP-1],1)
print("kernel %.2f ms, %d slices, mark %.1f"%(ms,n,mark))
tot=0; rows=[]
for k in range(NP-1):
    net=acc[k]-mark*cnt[k]; tot+=net; rows.append((k,pt.PHASES[k],net/mbs,cnt[k]/mbs))
for k,name,c,calls in rows:
    print("%2d %-58s %8.1f cyc/MB %5.1f%% %6.2f calls/MB"%(k,name,c,100*c*mbs/tot,calls))
print("sum %.1f"%(tot/mbs))
