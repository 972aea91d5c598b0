import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "tools"))
import config_sweep as CS, torch
CS.train_cfg("C4 whole batch on ONE GPU", 256, 16, 8192, True, 0., 1., iters=5)
print("peak mem GB", torch.cuda.max_memory_allocated() / 2**30, "(torch only; stash is hipMalloc)")
free, total = torch.cuda.mem_get_info(); print("device used GB", (total - free) / 2**30)
CS.train_cfg("N=32768 K=4", 256, 4, 32768, True, 0., 1., iters=3)
free, total = torch.cuda.mem_get_info(); print("device used GB", (total - free) / 2**30)
