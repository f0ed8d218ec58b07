import torch, time
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
dev = torch.device("cuda:0")
def t(f, n=20000):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("current_stream()        %.2f us" % t(lambda: torch.cuda.current_stream()))
print("current_stream(dev)     %.2f us" % t(lambda: torch.cuda.current_stream(dev)))
print("current_stream(0)       %.2f us" % t(lambda: torch.cuda.current_stream(0)))
print("is_available()          %.2f us" % t(lambda: torch.cuda.is_available()))
print("_cuda_getDeviceCount    %.2f us" % t(lambda: torch._C._cuda_getDeviceCount()))
s = torch.cuda.current_stream()
print(".cuda_stream            %.2f us" % t(lambda: s.cuda_stream))
print("is_current_stream_capturing %.2f us" % t(lambda: torch.cuda.is_current_stream_capturing()))
x = torch.zeros(4, device=dev)
print("tensor._version         %.2f us" % t(lambda: x._version))
