"""How does a replayed HIP graph with parallel branches start its branches?  Three chains of N tiny kernels (distinct
kernels per chain so the trace can tell them apart), forked from one stream and joined again; the graph is replayed
back to back with the host running ahead.  Run under `rocprofv3 --kernel-trace`; tools/micro/graph_branch_probe_read.py
prints, for the last replays, when each chain's FIRST and LAST kernel started relative to the replay's first kernel.
Also prints the host time of one replay() call."""
import sys, time, torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
order = sys.argv[2] if len(sys.argv) > 2 else 'abc'
NFILL = int(sys.argv[3]) if len(sys.argv) > 3 else 2
SERIAL = len(sys.argv) > 4 and sys.argv[4] == 'serial'
dev = torch.device('cuda')
a = torch.zeros(256, device=dev); b = torch.zeros(256, device=dev); c = torch.zeros(256, device=dev)
big = torch.zeros(64 << 20, device=dev)
main = torch.cuda.Stream(); s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
def chain(name):
    if name == 'a':
        for _ in range(N): a.add_(1.0)          # add kernel
    elif name == 'b':
        for _ in range(N): b.mul_(1.0001)       # mul kernel
    else:
        for _ in range(N): c.sin_()             # sin kernel
def body():
    ev = torch.cuda.Event(); ev.record()
    for ch in order:
        st = {'a': None, 'b': s1, 'c': s2}[ch] if not SERIAL else None
        if st is None:
            chain(ch)
        else:
            st.wait_event(ev)
            with torch.cuda.stream(st):
                chain(ch)
    cur = torch.cuda.current_stream()
    cur.wait_stream(s1); cur.wait_stream(s2)
    for _ in range(NFILL - 1):                  # ~ the long kernels of the step (let the host run ahead): 256 MB fills
        big.fill_(1.0)
    big.fill_(2.0)
with torch.cuda.stream(main):
    for _ in range(3): body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        body()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(('serial ' if SERIAL else 'forked ') + 'N=%d order=%s: host time per replay() %.1f us, total per replay incl. GPU %.1f us' % (N, order, (t1 - t0) / 20 * 1e6, (t2 - t0) / 20 * 1e6))
