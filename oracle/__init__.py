"""CPU oracle for the fused tabular-transformer + PNA hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product package (``models-for-relational-multimodal-data_amd/tabgnn_amd``) never
imports it and fails loudly when its HIP library is missing.

The oracle is a *functional* restatement (plain torch fp32 ops over a flat
``state_dict``) of the reference path, each function citing the reference
``file:line`` it follows:

=====================  =====================================================  ==========
module                 restates                                               parity pin
=====================  =====================================================  ==========
``transformer``        torch ``nn.TransformerEncoderLayer`` as configured at  pinned (torch itself, run here)
                       ``src/nn/models/fused.py:83-92,187-196``
``fused_path``         ``src/nn/models/fused.py:144-175,248-269``             pinned (reference file shim-imported,
                                                                              ``tests/golden/make_golden.py``)
``tabgnn_path``        ``src/nn/models/tabgnn.py:100-151,187-191,218-219``    pinned (same)
``heads``              ``src/nn/gnn/decoder.py:5-32``                         pinned (same)
``pna``                torch_geometric 2.5.3 ``PNAConv`` / ``BatchNorm`` /    **parity unpinned** (third-party source
                       ``DegreeScalerAggregation`` (``environment.yml:336``)  absent from /root/reference; restated
                       + ``src/nn/gnn/pna.py:17-46`` (PNAConvHetero, pinned)  from the published 2.5.3 algorithm)
``encoders``           Atahanak/pytorch-frame fork stype encoders             **parity unpinned** (un-vendored
                       (``.gitmodules:1-3``; call sites                       submodule; restated from upstream
                       ``src/datasets/ibm_transactions_for_aml.py:283-319``)  pytorch-frame 0.2.x semantics)
``step``               ``utils.py:353-362`` wrapper forward, ``main.py:52-75``  pinned through the pieces above
                       weighted CE + Adam
=====================  =====================================================  ==========
"""
