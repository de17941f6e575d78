"""Where a replayed step waits although the DAG does not ask for it.

    python tools/graph_node_waits.py step.dot kernel_trace.csv [min-wait-us]

Matches the kernel nodes of the captured graph (T2V_GRAPH_DOT, see graph_critical_path.py) to the kernel launches of the LAST
traced step — the k-th node of a (kernel, grid) pair in capture order is the k-th launch of that pair by start time — and prints,
for every node, start - max(end of its DAG predecessors): time the node was ready but not running (the graph executor's
node -> queue mapping, or no free CU slots), largest first, with the queue the launch was seen on."""
import collections, csv, re, subprocess, sys

dot_path, trace_path = sys.argv[1], sys.argv[2]
min_wait = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
txt = open(dot_path).read()
nodes = {}
for m in re.finditer(r'"graph_0_node_(\d+)"\[[^\]]*?label="\{\s*(\w+)(.*?)\}"\];', txt, re.S):
    nid, kind, body = int(m.group(1)), m.group(2), m.group(3)
    km = re.search(r'\|\s*\d+\s*\|\s*(\S+?)\\<\\<\\<\((\d+),(\d+),(\d+)\)', body)
    nodes[nid] = [km.group(1), int(km.group(2)) * int(km.group(3)) * int(km.group(4))] if km else [kind, 0]
edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_0_node_(\d+)" -> "graph_0_node_(\d+)"', txt)]
mangled = sorted({v[0] for v in nodes.values() if v[0].startswith('_Z')})
dem = subprocess.run(['c++filt'] + mangled, capture_output=True, text=True).stdout.splitlines()
dm = dict(zip(mangled, dem))
strip = lambda n: re.sub(r'\s+', '', n)
rows = []
with open(trace_path) as f:
    for r in csv.DictReader(f):
        n = 1
        for d in 'XYZ':
            n *= max(int(r['Grid_Size_' + d]) // max(int(r['Workgroup_Size_' + d]), 1), 1)
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), n))
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2].startswith('k_clip_adam')]
step = rows[adam[-2] + 1:adam[-1] + 1]
t0 = step[0][0]
bykey = collections.defaultdict(list)
for s, e, n, q, g in step:
    bykey[(strip(n), g)].append((s, e, q))
used = collections.defaultdict(int)
start, end, queue = {}, {}, {}
for nid in sorted(nodes):
    name, grid = nodes[nid]
    key = (strip(dm.get(name, name)), grid)
    k = used[key]
    if k < len(bykey.get(key, ())):
        s, e, q = bykey[key][k]
        start[nid], end[nid], queue[nid] = (s - t0) / 1e3, (e - t0) / 1e3, q
        used[key] += 1
pred = collections.defaultdict(list)
for a, b in edges:
    pred[b].append(a)
def ready(n, seen=None):          # end of the latest matched predecessor (unmatched nodes — memsets — are looked through)
    best = (0.0, None)
    for p in pred[n]:
        if p in end:
            best = max(best, (end[p], p))
        else:
            best = max(best, ready(p))
    return best
print("%d of %d nodes matched to launches of the last step (%.1f us long)" % (len(start), len(nodes), (step[-1][1] - t0) / 1e3))
out = []
for n in start:
    r, p = ready(n)
    out.append((start[n] - r, n, r, p))
print("%9s %9s %9s %5s  %-46s <- last predecessor" % ("wait us", "ready", "start", "queue", "node"))
for w, n, r, p in sorted(out, reverse=True):
    if w < min_wait:
        break
    pn = "#%d %s (q%s)" % (p, dm.get(nodes[p][0], nodes[p][0])[:36], queue.get(p)) if p is not None else "-"
    print("%9.1f %9.1f %9.1f %5s  #%-4d %-40s <- %s" % (w, r, start[n], queue[n], n, dm.get(nodes[n][0], nodes[n][0])[:40], pn))
print("sum of waits >= %.0f us: %.1f us" % (min_wait, sum(w for w, *_ in out if w >= min_wait)))
