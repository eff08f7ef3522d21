"""Downstream consumer of the pretrained encoder: the reference's ``Enhance_model.forward_pretrain`` +
``Fusion`` (model/Model.py:5-18, 20-46, 91-107) without the predictor zoo.  The frozen GPT-ST encoder (``mode='eval'``:
``dim_in_flow`` + one STHCN, no masking) runs on the HIP kernels; ``lin_test`` and the ``Fusion`` gate keep the reference's parameters
(state_dict keys) and run as one fused HIP launch (round 4, fusion.py) with a torch fallback, so any predictor (an ``nn.Module`` taking the (B,T,N,C)
embedding) can be trained on top with ordinary autograd."""
import torch
import torch.nn as nn

from .fusion import fusion_gate
from .model import GPTST_Model


class Fusion(nn.Module):                                                   # model/Model.py:5-18
    def __init__(self, dim):
        super().__init__()
        self.HS_fc = nn.Linear(dim, dim, bias=True)
        self.HT_fc = nn.Linear(dim, dim, bias=True)
        self.output_fc = nn.Linear(dim, dim, bias=True)

    def forward(self, flow_eb, time_eb):
        z = torch.sigmoid(self.HS_fc(flow_eb) + self.HT_fc(time_eb))
        return self.output_fc(z * flow_eb + (1 - z) * time_eb)


class EnhanceFrontEnd(nn.Module):
    """``Enhance_model`` in ``mode='eval'`` (model/Model.py:40-46,91-107): frozen pretrained encoder -> fusion with a linear lift
    of the raw flow -> predictor.  ``predictor=None`` returns the fused embedding."""

    def __init__(self, args, predictor=None):
        super().__init__()
        assert args.mode == "eval", "the enhanced front end wraps the encoder in eval mode (reference Run.py -mode eval)"
        self.input_base_dim = args.input_base_dim
        self.pretrain_model = GPTST_Model(args)
        for p in self.pretrain_model.parameters():                         # :93-94
            p.requires_grad = False
        self.fusion = Fusion(args.hidden_dim)
        self.lin_test = nn.Linear(args.input_base_dim, args.hidden_dim)
        self.predictor = predictor

    def load_pretrained_model(self, path_or_state):                        # :91-92
        sd = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, str) else path_or_state
        self.pretrain_model.load_state_dict(sd)

    def forward(self, source, label=None, batch_seen=None):                # :96-107
        x_pretrain_flow = self.pretrain_model(source, label)[0]
        # :105-107 lin_test + Fusion: one HIP launch forward (csrc/fusion.hip) where the shape allows, the torch modules otherwise
        eb = fusion_gate(x_pretrain_flow.detach(), source, self.fusion, self.lin_test, self.input_base_dim)
        if self.predictor is None:
            return eb
        x = self.predictor(eb)
        return x, x, x, x, x
