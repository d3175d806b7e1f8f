"""eeg_gnn_ssl_amd — MI355X-native DCRNN forward/backward for the 19-electrode EEG graph.

Drop-in for the DCRNN hot path of tsy935/eeg-gnn-ssl (model/cell.py, model/model.py): the same
`nn.Module` classes, signatures and `state_dict` layout, computed by hand-written HIP kernels for
gfx950 behind a C ABI (include/eeg_dcrnn.h). 
"""
from . import ops, utils                                  # noqa: F401
from .model.cell import DCGRUCell, DiffusionGraphConv     # noqa: F401
from .model.model import (DCGRUDecoder, DCRNNEncoder, DCRNNModel_classification,   # noqa: F401
                          DCRNNModel_nextTimePred)

__all__ = ["DCGRUCell", "DiffusionGraphConv", "DCRNNEncoder", "DCGRUDecoder",
           "DCRNNModel_classification", "DCRNNModel_nextTimePred", "ops", "utils"]
