import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "models-for-relational-multimodal-data_amd"))
import torch, tabgnn_amd as T
from tabgnn_amd import synthetic as S
st = T.stype
dev = "cuda:0"
R = 430000
num, cat, ts = S.edge_table(R, 0)
def run(cols, stats, feats, label):
    enc = T.StypeWiseFeatureEncoder(128, stats, cols, torch.bfloat16).to(dev)
    tf = T.TensorFrame({k: torch.from_numpy(v).to(dev) for k, v in feats.items()}, cols)
    out, _ = enc(tf)
    g = torch.randn_like(out)
    for _ in range(2):
        out, _ = enc(tf); out.backward(g)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(); out, _ = enc(tf); e1.record(); out.backward(g); e2.record(); torch.cuda.synchronize()
    print(f"{label:28s} fwd {e0.elapsed_time(e1)*1e3:8.1f} us  bwd {e1.elapsed_time(e2)*1e3:8.1f} us")
run({st.numerical: ["Amount Paid"]}, S.EDGE_STATS, {st.numerical: num}, "num only")
run({st.categorical: ["Payment Currency"]}, S.EDGE_STATS, {st.categorical: cat[:, :1].copy()}, "1 cat")
run({st.categorical: S.EDGE_COLS[st.categorical]}, S.EDGE_STATS, {st.categorical: cat}, "3 cat")
run({st.timestamp: ["Timestamp"]}, S.EDGE_STATS, {st.timestamp: ts}, "ts only")
run(S.EDGE_COLS, S.EDGE_STATS, {st.numerical: num, st.categorical: cat, st.timestamp: ts}, "all 5")
