"""tabgnn_amd — MI355X-native drop-in for the fused tabular-transformer + PNA hot path of
Atahanak/models-for-relational-multimodal-data (``src/nn/models/fused.py`` behind ``utils.py:TABGNNFusedS``).

Importing this package loads ``libtabgnn_hip.so`` (C ABI: ``include/tabgnn_hip.h``); it raises if the library
is missing.  There is no CPU/eager fallback: tensors must live on a gfx950 device.
"""
from . import _lib

_lib.load()

from .encoders import (EmbeddingEncoder, LinearEncoder, ProjectionEncoder,  # noqa: E402
                       StypeWiseFeatureEncoder, TimestampEncoder)
from .frame import TensorFrame, stype  # noqa: E402
from .heads import ClassifierHead, LinkPredHead, MCMHead, NodeClassificationHead, SelfSupervisedHead  # noqa: E402
from .losses import SSLoss  # noqa: E402
from .layers import BatchNorm, ColumnTransformerLayer, GINEConv, GINEConvHetero, PNAConv, PNAConvHetero  # noqa: E402
from .models import (CPNA, PNAS, TABGNN, GINe, FTTransformerLayer, FTTransformerPNAFusedLayer,  # noqa: E402
                     FTTransformerPNAInterleavedLayer, PNALayer, TABGNNFused, TABGNNInterleaved)
from .train import DataParallel, FlatParams, FusedAdam, IndexGuard, train_step  # noqa: E402
from .graph_step import GraphedTrainStep, StepState, bucket_size, prepare as prepare_batch  # noqa: E402
from .wrappers import GNN, TABGNNFusedS, TABGNNS, degree_histogram  # noqa: E402
from .device_sampler import (DeviceBatchLoader, DeviceNeighborSampler, device_batch_index,  # noqa: E402
                             prepare_sample_device)
