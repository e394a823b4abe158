import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prior_flow_amd import det_state_dict, synthetic_pair
from prior_flow_amd.modules import state_dict_shapes
from prior_flow_amd.prior_raft import PriOr_RAFT
params = det_state_dict(state_dict_shapes())
def model(streams, graph):
    m = PriOr_RAFT(argparse.Namespace(mixed_precision=False, dropout=0.0)); m.load_state_dict(params, strict=True)
    m = m.cuda().eval(); m.use_streams, m.use_graph = streams, graph; return m
H, W = int(os.environ.get('H', 128)), int(os.environ.get('W', 256))
i1, i2 = synthetic_pair(1, H, W); i1, i2 = i1.cuda(), i2.cuda()
with torch.no_grad():
    for streams, graph in ((True, True),):
        m = model(streams, graph)
        outs = [m(i1, i2, iters=12, test_mode=True).clone() for _ in range(int(os.environ.get("RUNS", "12")))]
        d = [float((o - outs[0]).abs().max()) for o in outs]
        print(f'{H}x{W}', 'forks', os.environ.get('PRIORFLOW_FORKS'), sum(x != 0 for x in d), 'of', len(d), 'differ', max(d))
