"""DeepFM (drop-in for ``torch_rechub.models.ranking.DeepFM``,
/root/reference/recbox/third_party/rechub/models/ranking/deepfm.py:14-42)."""
import torch

from ... import dense, ops
from ..basic.features import DenseFeature, SparseFeature
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

    def _shared_gather(self):
        """True when the deep input is [the FM embeddings flattened | dense values]: the sparse features of
        ``deep_features`` are exactly ``fm_features`` in the same order (how deepfm.py is used with Criteo)."""
        sparse = [f for f in self.deep_features if not isinstance(f, DenseFeature)]
        return (len(sparse) == len(self.fm_features) and all(a is b for a, b in zip(sparse, self.fm_features))
                and len(set(f.embed_dim for f in self.fm_features)) == 1
                and all(isinstance(f, SparseFeature) for f in self.fm_features))

    def forward(self, x):
        if self._shared_gather():
            # deepfm.py:34-35 looks every table up twice (deep input, FM input) and autograd then adds two dense
            # [V, D] gradients per table.  Here ONE gather produces [B, F*D | dense] (rows padded to 16 bytes); the
            # FM part and the LR part read its leading F*D columns in place, the tower reads the whole row.
            input_deep = self.embedding(x, self.deep_features, squeeze_dim=True)
            dim = self.fm_features[0].embed_dim
            mods = self.mlp.mlp
            if (ops.config.fuse_deepfm_input and len(mods) > 0 and type(mods[0]) is torch.nn.Linear
                    and not self.linear.sigmoid and ops.deepfm_input_stage_supported(input_deep, self.fm_dims, dim)):
                # the three readers of the block as ONE autograd node: its gradient comes out of the tower's dx GEMM
                h, y_fm, y_linear = ops.deepfm_input_stage(input_deep, mods[0], self.linear.fc, self.fm_dims, dim)
                y_deep = dense.run_sequential(mods[1:], h)
                return ops.sigmoid_output((y_linear + y_fm + y_deep).squeeze(1))
            if input_deep.is_cuda and input_deep.dim() == 2:
                # three readers of one block: its gradient is assembled by one kernel (ops.shared_prefix)
                input_deep, flat_fm, flat_lr = ops.shared_prefix(input_deep, self.fm_dims, copies=2)
            else:
                flat_fm = flat_lr = input_deep[:, :self.fm_dims]
            input_fm = flat_fm.view(flat_fm.shape[0], len(self.fm_features), self.fm_features[0].embed_dim)
            y_linear = self.linear(flat_lr)
        else:
            input_deep = self.embedding(x, self.deep_features, squeeze_dim=True)     # [B, deep_dims]
            input_fm = self.embedding(x, self.fm_features, squeeze_dim=False)        # [B, F, D]
            y_linear = self.linear(input_fm.flatten(start_dim=1))
        y_fm = self.fm(input_fm)
        y_deep = self.mlp(input_deep)
        y = y_linear + y_fm + y_deep
        return ops.sigmoid_output(y.squeeze(1))
