import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import mp3rgain_amd as rg
from mp3rgain_amd import _capi, album as album_mod
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
an = rg.Analyzer(0); an.set_stream(torch.cuda.current_stream().cuda_stream)
rate, frames = 44100, 44100*600
pcm = torch.empty((1,2,frames), dtype=torch.float32, device="cuda")
d = (_capi.TrackDesc*1)()
for c in range(2): an.synth_fill_device(pcm[0,c].data_ptr(), 1, c, rate, 0, frames)
d[0].offset_bytes, d[0].frames, d[0].sample_rate, d[0].channels, d[0].format = 0, frames, rate, 2, 0
class DA:
    def __init__(s,p,sh,t): s.__cuda_array_interface__={"shape":sh,"typestr":t,"data":(p,False),"version":3}
views={}
T={"enq":0,"view":0,"ar":0,"res":0}
def step(timed):
    t0=time.perf_counter(); an.enqueue_device(d,1,pcm.data_ptr(),pcm.numel()*4,album=True); t1=time.perf_counter()
    v=an.device_view()
    if v.d_album_hist not in views: views[v.d_album_hist]=(torch.as_tensor(DA(v.d_album_hist,(12000,),"<i4"),device="cuda"),torch.as_tensor(DA(v.d_album_peak,(1,),"<f8"),device="cuda"))
    h,p=views[v.d_album_hist]; t2=time.perf_counter()
    album_mod.allreduce_album(h,p,even_if_alone=True); t3=time.perf_counter()
    an.album_result_enqueue(); t4=time.perf_counter()
    if timed:
        T["enq"]+=t1-t0; T["view"]+=t2-t1; T["ar"]+=t3-t2; T["res"]+=t4-t3
for _ in range(20): step(False)
torch.cuda.synchronize()
K=200; t0=time.perf_counter()
for _ in range(K): step(True)
th=time.perf_counter()-t0; torch.cuda.synchronize(); tt=time.perf_counter()-t0
print({k:round(v/K*1e6,1) for k,v in T.items()}, "host us/step", round(th/K*1e6,1), "total us/step", round(tt/K*1e6,1))
an.close(); dist.destroy_process_group()
