"""Round 5: is a hipMemsetAsync captured into a hipGraph executed on replay?  (The library clears its amax slots with one per call; a graph whose
memset were dropped would keep the atomicMax of every earlier replay in its slots -- invisible while a test replays the data it captured with.)"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for nbytes in (4096, 64 * 512 * 128 * 4):
    n = nbytes // 4
    buf = torch.full((n,), 7.0, device="cuda")
    x = torch.ones(n, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
            buf.add_(x)
    vals = []
    for _ in range(3):
        g.replay(); torch.cuda.synchronize(); vals.append((float(buf[0]), float(buf[-1]), float(buf.sum()) / n))
    print(f"memset of {nbytes} bytes in a graph, rc {rc}: after replays 1..3 buf = {vals} (1.0 everywhere = the memset node runs)")
