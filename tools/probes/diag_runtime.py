import sys, importlib, ctypes, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    api = importlib.import_module("3dbodyanimation_amd.api")
    print("lib count", api.device_count())
    x = torch.zeros(4, device="cuda"); print("tensor ok", x.sum().item())
else:
    api = importlib.import_module("3dbodyanimation_amd.api")
    print("lib count", api.device_count())
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
os.system("grep -E 'amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
