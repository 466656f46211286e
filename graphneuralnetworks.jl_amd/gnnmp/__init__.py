"""gnnmp — MI355X-native message-passing engine behind the GNNlib.jl / GNNGraphs.jl API for the
propagate / apply_edges / aggregate_neighbors hot path (include/gnnmp.h).

Host side only: every arithmetic step is a call into libgnnmp.so (hand-written HIP for gfx950).
"""
from ._lib import GnnmpError, knob, load, tune  # noqa: F401
from . import placement  # noqa: F401
from .graph import (GNNGraph, Plan, add_self_loops, batch, batch_arrays, check_num_edges, check_num_nodes, degree,  # noqa: F401
                    edge_index, get_edge_weight, get_graph_type, graph_indicator, set_edge_weight)
from .layers import (Dense, GATConv, GCNConv, GlobalPool, GNNChain, GraphConv, SAGEConv, bias_act, dense, fused_conv,  # noqa: F401
                     gat_conv, gcn_conv, global_pool, glorot_uniform, graph_conv, sage_conv)
from .msgpass import (aggregate_neighbors, apply_edges, copy_xi, copy_xj, e_mul_xj, propagate, w_mul_xj,  # noqa: F401
                      xi_dot_xj, xi_sub_xj, xj_sub_xi)
from .layers_attn import (AGNNConv, GATv2Conv, GINConv, TransformerConv, agnn_conv, gatv2_conv, gin_conv,  # noqa: F401
                          transformer_conv)
from .layers_khop import (GlobalAttentionPool, ResGatedGraphConv, SGConv, TAGConv, global_attention_pool,  # noqa: F401
                          res_gated_graph_conv, sg_conv, tag_conv)
from .layers_more import (CGConv, ChebConv, DConv, EdgeConv, EGNNConv, GatedGraphConv, GMMConv, MEGNetConv,  # noqa: F401
                          NNConv, Set2Set, cg_conv, cheb_conv, d_conv, edge_conv, egnn_conv, gated_graph_conv, gmm_conv,
                          megnet_conv, nn_conv, set2set_pool)
from .dataset import DataLoader, GraphDataset, concat_plans  # noqa: F401
from .sampling import (NeighborLoader, NodeSet, has_self_loops, induced_subgraph, is_bidirected, sample_neighbors,  # noqa: F401
                       sort_edge_index)
from .utils import (broadcast_edges, broadcast_nodes, expand_srcdst, reduce_edges, reduce_nodes,  # noqa: F401
                    softmax_edge_neighbors, softmax_edges, softmax_nodes)

__version__ = "0.1.0"
