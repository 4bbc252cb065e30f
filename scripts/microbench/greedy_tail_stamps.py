import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import gobblet_rl_amd as G

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])
boards=65536
nat, L = G._native, G._native.lib()
env = G.BatchedGobblet(boards, "cuda:0", auto_reset=True, seed=0); env.rollout(64)
act = torch.empty(boards, dtype=torch.int32, device="cuda:0"); cm = torch.empty((boards, 54), dtype=torch.int8, device="cuda:0"); fb = torch.empty(boards, dtype=torch.int8, device="cuda:0")
for _ in range(3):
    nat.check(L.gbl_greedy(env.squares.data_ptr(), env.to_move.data_ptr(), None, None, 2, act.data_ptr(), cm.data_ptr(), fb.data_ptr(), boards, None))
torch.cuda.synchronize()
buf = np.zeros((1024, 16, 12), np.uint64)
L.gbl_debug_wave_stamps.argtypes = [C.c_void_p]
assert L.gbl_debug_wave_stamps(buf.ctypes.data) == 0
t = buf.astype(np.int64); t = t[t[:,0,0]>0]
o = t[:, :4, :]
print("owners (mean cycles): past E -> merged %.0f | merged -> replayed %.0f | replayed -> outputs issued %.0f | total tail %.0f" % (
    (o[:,:,6]-o[:,:,5]).mean(), (o[:,:,7]-o[:,:,6]).mean(), (o[:,:,11]-o[:,:,7]).mean(), (o[:,:,11]-o[:,:,5]).mean()))
