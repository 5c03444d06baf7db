import ctypes as C, os, sys
mode = sys.argv[1]
if mode == "torch_first":
    import torch; print("torch avail", torch.cuda.is_available())
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spike-petsc_amd", "libspike_mi355.so"))
hip = None
for line in open("/proc/self/maps"):
    if "amdhip64" in line or "hsa-runtime" in line:
        p = line.split()[-1]
        if p != hip: print("mapped:", p); hip = p
h = C.c_void_p()
print("spike_create rc", L.spike_create(C.byref(h)))
H = C.CDLL("libamdhip64.so.7")
n = C.c_int(-1)
H.hipGetErrorString.restype = C.c_char_p
rc = H.hipGetDeviceCount(C.byref(n)); print("hipGetDeviceCount", rc, n.value, H.hipGetErrorString(rc))
if mode != "torch_first":
    import torch; print("torch avail", torch.cuda.is_available())
