import sys, time, cProfile, pstats, io, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from oracle import fake_diffusers as fd
import test_gpu_integration as T
import daam_amd
pipe = fd.make_pipe('sdxl', device='cuda:0', dtype=torch.float16, batch=2, seed=3, mini=False, identity_proj=False)
T._resident_inputs(pipe, 4)
mods = [s.module for s in pipe.unet.execution_order()]
for m in mods: m.set_processor(T._SdpaProcessor())
prompt='a photo of a monkey'
def sync(): torch.cuda.synchronize()
# 1. trace setup/teardown only
for _ in range(2):
    with daam_amd.trace(pipe) as tc: pass
sync(); t0=time.perf_counter()
for _ in range(5):
    with daam_amd.trace(pipe) as tc: pass
sync(); print('trace enter+exit (no generation): %.2f ms' % ((time.perf_counter()-t0)/5*1e3))
# 2. generation inside one long-lived trace
with daam_amd.trace(pipe) as tc:
    pipe(prompt, num_inference_steps=5); tc.compute_global_heat_map(); sync()
    t0=time.perf_counter(); pipe(prompt, num_inference_steps=20); sync(); t1=time.perf_counter()
    tc.compute_global_heat_map(); sync(); t2=time.perf_counter()
    print('traced 20 steps: %.3f ms/step; compute_global_heat_map (incl. tap launch): %.2f ms' % ((t1-t0)/20*1e3, (t2-t1)*1e3))
    pr=cProfile.Profile(); pr.enable(); pipe(prompt, num_inference_steps=5); sync(); pr.disable()
    s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:2600])
sync(); t0=time.perf_counter(); pipe(prompt, num_inference_steps=20); sync(); print('plain 20 steps: %.3f ms/step' % ((time.perf_counter()-t0)/20*1e3))
