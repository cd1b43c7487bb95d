"""DeepFM (drop-in for ``torch_rechub.models.ranking.DeepFM``,
/root/reference/recbox/third_party/rechub/models/ranking/deepfm.py:14-42)."""
import torch

from ..basic.layers import FM, LR, MLP, EmbeddingLayer


class DeepFM(torch.nn.Module):
    def __init__(self, deep_features, fm_features, mlp_params):
        super(DeepFM, self).__init__()
        self.deep_features = deep_features
        self.fm_features = fm_features
        self.deep_dims = sum([fea.embed_dim for fea in deep_features])
        self.fm_dims = sum([fea.embed_dim for fea in fm_features])
        self.linear = LR(self.fm_dims)          # first order: ONE Linear over the flattened embeddings
        self.fm = FM(reduce_sum=True)           # second order
        self.embedding = EmbeddingLayer(deep_features + fm_features)
        self.mlp = MLP(self.deep_dims, **mlp_params)

    def forward(self, x):
        input_deep = self.embedding(x, self.deep_features, squeeze_dim=True)     # [B, deep_dims]
        input_fm = self.embedding(x, self.fm_features, squeeze_dim=False)        # [B, F, D]
        y_linear = self.linear(input_fm.flatten(start_dim=1))
        y_fm = self.fm(input_fm)
        y_deep = self.mlp(input_deep)
        y = y_linear + y_fm + y_deep
        return torch.sigmoid(y.squeeze(1))
