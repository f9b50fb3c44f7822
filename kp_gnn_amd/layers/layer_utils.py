"""Layer factory, same contract as the reference's layers/layer_utils.py:10-34."""
from .KPGCN import KPGCNConv
from .KPGIN import KPGINConv
from .KPGINplus import KPGINPlusConv


def make_gnn_layer(args):
    """One layer (cloned by the body) for KPGCN / KPGIN / KPGINPrime; a LIST of num_layer layers with
    K_l = min(l, K) for KPGINPlus."""
    name = args.model_name
    common = dict(num_hop1_edge=args.num_hop1_edge, num_pe=args.max_pe_num, combine=args.combine)
    if name == "KPGCN":
        return KPGCNConv(args.hidden_size, args.hidden_size, args.K, **common)
    if name in ("KPGIN", "KPGINPrime"):
        return KPGINConv(args.hidden_size, args.hidden_size, args.K, eps=args.eps, train_eps=args.train_eps, **common)
    if name == "KPGINPlus":
        return [KPGINPlusConv(args.hidden_size, args.hidden_size, min(l, args.K), **common)
                for l in range(1, args.num_layer + 1)]
    if name == "KPGraphSAGE":
        from .KPGraphSAGE import KPGraphSAGEConv
        return KPGraphSAGEConv(args.hidden_size, args.hidden_size, args.K, args.aggr, **common)
    raise ValueError("Not supported GNN type")
