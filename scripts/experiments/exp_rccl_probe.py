"""probe: does RCCL initialise on this box — torch's bundled copy and the system copy through our C ABI"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
which = sys.argv[1]
if which == "torch":
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:29511", device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("TORCH_NCCL_OK", t.tolist())
else:
    if which == "preload":
        ctypes.CDLL(sys.argv[2], mode=ctypes.RTLD_GLOBAL)
    from xsdba_amd import hip
    from xsdba_amd.sharding import init_comm, comm_info
    ctx = hip.Context(0)
    init_comm(ctx, 0, 1)
    print("NATIVE_OK", comm_info(ctx))
