import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.version.hip)
    import raytracer_amd
    L = raytracer_amd.load_library()
    cfg = raytracer_amd.Config(); L.rt_hip_default_config(cfg)
    import ctypes as C
    h = C.c_void_p(); print("create", L.rt_hip_create(cfg, C.byref(h)), L.rt_hip_last_error())
    x = torch.zeros(10, device="cuda"); print(x.sum().item())
else:
    import raytracer_amd, ctypes as C
    L = raytracer_amd.load_library()
    cfg = raytracer_amd.Config(); L.rt_hip_default_config(cfg)
    h = C.c_void_p(); print("create", L.rt_hip_create(cfg, C.byref(h)), L.rt_hip_last_error())
    import torch
    print("torch avail", torch.cuda.is_available())
os.system("grep -i hip /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
