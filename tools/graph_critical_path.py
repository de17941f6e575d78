"""Critical path of the captured training step.

    T2V_GRAPH_DOT=step.dot python bench.py --bf16 ...      # train.py dumps the captured DAG (hipGraphDebugDotPrint)
    python tools/graph_critical_path.py step.dot kernel_trace.csv [from-kernel]

Nodes = the kernel / memset nodes of the graph, edges = its dependencies (stream order and event waits as captured), node weight =
the mean traced duration of that (kernel, grid) in a rocprofv3 kernel trace of the same configuration.  Prints the longest
weighted path (the bound a perfect scheduler with infinite CUs could reach), the slack of every heavy node, and — with
`from-kernel` — the longest path that starts at that kernel (e.g. the tail behind the reverse pass)."""
import collections, csv, re, subprocess, sys

dot_path, trace_path = sys.argv[1], sys.argv[2]
start_from = sys.argv[3] if len(sys.argv) > 3 else None
txt = open(dot_path).read()
nodes = {}
for m in re.finditer(r'"graph_0_node_(\d+)"\[[^\]]*?label="\{\s*(\w+)(.*?)\}"\];', txt, re.S):
    nid, kind, body = int(m.group(1)), m.group(2), m.group(3)
    name, grid = kind, 0
    km = re.search(r'\|\s*\d+\s*\|\s*(\S+?)\\<\\<\\<\((\d+),(\d+),(\d+)\),\((\d+),(\d+),(\d+)\)', body)
    if km:
        name = km.group(1)
        grid = int(km.group(2)) * int(km.group(3)) * int(km.group(4))
    nodes[nid] = [name, grid]
edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_0_node_(\d+)" -> "graph_0_node_(\d+)"', txt)]
mangled = sorted({v[0] for v in nodes.values() if v[0].startswith('_Z')})
dem = subprocess.run(['c++filt'] + mangled, capture_output=True, text=True).stdout.splitlines()
dm = dict(zip(mangled, dem))
for v in nodes.values():
    v[0] = dm.get(v[0], v[0])

# mean duration per (name, workgroups) over the trace (all steps; the first replays included — shapes do not change)
acc = collections.defaultdict(lambda: [0, 0])
with open(trace_path) as f:
    for r in csv.DictReader(f):
        wg = [int(r['Workgroup_Size_' + d]) for d in 'XYZ']
        gs = [int(r['Grid_Size_' + d]) for d in 'XYZ']
        n = 1
        for g, w in zip(gs, wg):
            n *= max(g // max(w, 1), 1)
        a = acc[(r['Kernel_Name'], n)]
        a[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a[1] += 1
def strip(n):
    return re.sub(r'\s+', '', n)
by = {(strip(k[0]), k[1]): v[0] / v[1] / 1e3 for k, v in acc.items()}
byname = collections.defaultdict(list)
for (n, g), d in by.items():
    byname[n].append(d)
w = {}
missing = 0
for nid, (name, grid) in nodes.items():
    key = (strip(name), grid)
    if key in by:
        w[nid] = by[key]
    elif strip(name) in byname:
        w[nid] = sum(byname[strip(name)]) / len(byname[strip(name)])
    else:
        w[nid] = 3.0 if name in ('MEMSET', 'MEMCPY') else 5.0
        missing += name not in ('MEMSET', 'MEMCPY', 'EMPTY')
succ, pred = collections.defaultdict(list), collections.defaultdict(list)
for a, b in edges:
    succ[a].append(b)
    pred[b].append(a)
order, indeg = [], {n: len(pred[n]) for n in nodes}
q = [n for n in nodes if indeg[n] == 0]
while q:
    n = q.pop()
    order.append(n)
    for m_ in succ[n]:
        indeg[m_] -= 1
        if indeg[m_] == 0:
            q.append(m_)
est, best = {}, {}
for n in order:                         # earliest start / finish
    est[n] = max([est[p] + w[p] for p in pred[n]], default=0.0)
total = max(est[n] + w[n] for n in nodes)
lst = {}
for n in reversed(order):               # latest finish without stretching the path
    lst[n] = min([lst[s] - w[s] for s in succ[n]], default=total)
print("%d nodes (%d kernels), %d edges, %d kernel nodes without a traced duration; sum of durations %.1f us; critical path %.1f us" % (
    len(nodes), sum(1 for v in nodes.values() if v[0] not in ('MEMSET', 'MEMCPY', 'EMPTY')), len(edges), missing, sum(w.values()), total))

def path_to(end):
    p = [end]
    while pred[p[-1]]:
        p.append(max(pred[p[-1]], key=lambda x: est[x] + w[x]))
    return p[::-1]
end = max(nodes, key=lambda n: est[n] + w[n])
print("\ncritical path (earliest start us, duration us, grid, kernel):")
for n in path_to(end):
    if w[n] >= 8.0:
        print("  %9.1f %8.1f %6d  #%-4d %s" % (est[n], w[n], nodes[n][1], n, nodes[n][0][:90]))
print("\nheavy nodes OFF the critical path (duration >= 60 us) and their slack:")
for n in sorted(nodes, key=lambda n: est[n]):
    slack = lst[n] - (est[n] + w[n])
    if w[n] >= 60.0 and slack > 1.0:
        print("  %9.1f %8.1f  slack %8.1f  #%-4d %s" % (est[n], w[n], slack, n, nodes[n][0][:80]))
if start_from:
    cand = [n for n in nodes if nodes[n][0].startswith(start_from)]
    if cand:
        s0 = cand[-1]
        dist = {s0: 0.0}
        back = {}
        for n in order:
            if n in dist:
                for m_ in succ[n]:
                    d = dist[n] + w[n]
                    if d > dist.get(m_, -1.0):
                        dist[m_] = d
                        back[m_] = n
        e2 = max(dist, key=lambda n: dist[n] + w[n])
        p = [e2]
        while p[-1] in back:
            p.append(back[p[-1]])
        print("\nlongest path behind %s (#%d): %.1f us after its END" % (start_from, s0, dist[e2] + w[e2] - w[s0]))
        for n in p[::-1][1:]:
            if w[n] >= 5.0:
                print("  %9.1f %8.1f %6d  #%-4d %s" % (dist[n] - w[s0], w[n], nodes[n][1], n, nodes[n][0][:90]))
