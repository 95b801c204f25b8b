import time, numpy as np, torch, sys
sys.path.insert(0, '.')
import swarmmap_amd
from swarmmap_amd import synth
import bench
w,h=752,480
stream = synth.FrameStream(seed=3)
ex = swarmmap_amd.ORBextractor(1000,1.2,8,20,7)
frames=[stream.frame(t) for t in range(40)]
dev=[torch.from_numpy(f).cuda() for f in frames]
wl=bench.TrackingWorkload(stream,(w,h),7)
k,d=ex.run_device(dev[0].data_ptr(),w,h,w); wl.push(0,k,d)
m2=swarmmap_amd.ORBmatcher(0.9,True); m1=swarmmap_amd.ORBmatcher(0.8,True)
T=dict(ex=0,fv=0,q=0,m2=0,m1=0,k=0)
S2=dict(enqueue_ms=0,wait_ms=0,launches=0,staged_bytes=0); S1=dict(S2)
for t in range(1,240):
    a=time.perf_counter(); kps,desc=ex.run_device(dev[t%40].data_ptr(),w,h,w)
    b=time.perf_counter(); F=wl.frame_view(kps,desc)
    c=time.perf_counter(); last,mps=wl.queries(t)
    d_=time.perf_counter(); m2.SearchByProjectionLastFrame(F,last,15.0); k2=m2.last_kernel_ms(); s2=m2.last_stats()
    e=time.perf_counter(); m1.SearchByProjectionMapPoints(F,mps,1.0); k1=m1.last_kernel_ms(); s1=m1.last_stats()
    f=time.perf_counter(); wl.push(t,kps,desc)
    if t>=40:
        T['ex']+=b-a;T['fv']+=c-b;T['q']+=d_-c;T['m2']+=e-d_;T['m1']+=f-e;T['k']+=(k1+k2)*1e-3
        for k in S2: S2[k]+=s2[k]/200; S1[k]+=s1[k]/200
print({k:round(v/200*1e3,4) for k,v in T.items()}, len(mps['proj_x']), len(last['u']))
print('m2',S2); print('m1',S1)
# --- same loop with a local-mapping thread running LBA back to back
import threading
ba = swarmmap_amd.Optimizer()
win = synth.make_ba_case("LBA-M", seed=100)
stop_flag = [False]
def lm_loop():
    while not stop_flag[0]:
        ba.LocalBundleAdjustment(win)
th = threading.Thread(target=lm_loop, daemon=True); th.start()
T=dict(ex=0,fv=0,m2=0,m1=0,k=0); S2=dict(enqueue_ms=0,wait_ms=0,launches=0,staged_bytes=0); S1=dict(S2)
wl=bench.TrackingWorkload(stream,(w,h),7)
k,d=ex.run_device(dev[0].data_ptr(),w,h,w); wl.push(0,k,d)
for t in range(1,240):
    a=time.perf_counter(); kps,desc=ex.run_device(dev[t%40].data_ptr(),w,h,w)
    b=time.perf_counter(); F=wl.frame_view(kps,desc)
    c=time.perf_counter(); last,mps=wl.queries(t)
    d_=time.perf_counter(); m2.SearchByProjectionLastFrame(F,last,15.0); k2=m2.last_kernel_ms(); s2=m2.last_stats()
    e=time.perf_counter(); m1.SearchByProjectionMapPoints(F,mps,1.0); k1=m1.last_kernel_ms(); s1=m1.last_stats()
    f=time.perf_counter(); wl.push(t,kps,desc)
    if t>=40:
        T['ex']+=b-a;T['fv']+=c-b;T['m2']+=e-d_;T['m1']+=f-e;T['k']+=(k1+k2)*1e-3
        for k in S2: S2[k]+=s2[k]/200; S1[k]+=s1[k]/200
stop_flag[0]=True; th.join()
print('with LBA thread', {k:round(v/200*1e3,4) for k,v in T.items()})
print('m2',S2); print('m1',S1)
