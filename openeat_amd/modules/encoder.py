"""Encoder stack (/root/reference/openeat/modules/encoder.py:113-229):
[GlobalCMVN] -> Conv2dSubsampling4 (+pos. encoding) -> N x EncoderLayer -> LayerNorm(1e-5)."""
import torch

from openeat_amd import ops
from openeat_amd.modules.attention import MultiHeadedAttention, RelPositionMultiHeadedAttention
from openeat_amd.modules.convolution import ConvolutionModule
from openeat_amd.modules.embedding import PositionalEncoding, RelPositionalEncoding
from openeat_amd.modules.encoder_layer import EncoderLayer
from openeat_amd.modules.positionwise_feed_forward import PositionwiseFeedForward
from openeat_amd.modules.subsampling import (Conv2dSubsampling4, Conv2dSubsampling6, Conv2dSubsampling8,
                                             LinearNoSubsampling)
from openeat_amd.utils.common import get_activation


def _layers(d_model, dropout_rate, attention_heads, linear_units, activation_type, macaron_style, use_cnn_module,
            cnn_module_kernel, causal, use_adapter, n_unique, down_size=64, scalar=0.1):
    from openeat_amd.modules.adapter import Adapter
    attn_cls = RelPositionMultiHeadedAttention if use_cnn_module else MultiHeadedAttention

    def ff():
        return PositionwiseFeedForward(d_model, linear_units, dropout_rate, get_activation(activation_type))

    return torch.nn.ModuleList([
        EncoderLayer(d_model, ff() if macaron_style else None, attn_cls(attention_heads, d_model, dropout_rate),
                     ConvolutionModule(d_model, cnn_module_kernel, get_activation(activation_type), causal)
                     if use_cnn_module else None, ff(),
                     Adapter(d_model, dropout_rate, down_size, scalar) if use_adapter else None, dropout_rate)
        for _ in range(n_unique)])


class Encoder(torch.nn.Module):
    """encoder.py:25-110: the embedding-free stack (used by the LM in the reference)."""

    def __init__(self, d_model: int = 256, dropout_rate: float = 0.1, attention_heads: int = 4, linear_units: int = 2048,
                 activation_type: str = "swish", macaron_style: bool = True, use_cnn_module: bool = True,
                 cnn_module_kernel: int = 15, causal: bool = False, use_adapter: bool = False, down_size: int = 64,
                 scalar: float = 0.1, num_blocks: int = 6, num_blocks_share: int = 1):
        super().__init__()
        self._output_size = d_model
        self.num_blocks_share = num_blocks_share
        self.encoders = _layers(d_model, dropout_rate, attention_heads, linear_units, activation_type, macaron_style,
                                use_cnn_module, cnn_module_kernel, causal, use_adapter, num_blocks // num_blocks_share,
                                down_size, scalar)
        self.after_norm = torch.nn.LayerNorm(d_model, eps=1e-5)

    def output_size(self) -> int:
        return self._output_size

    def forward(self, xs: torch.Tensor, masks: torch.Tensor, pos_emb: torch.Tensor):
        m8 = ops.mask_bytes(masks)
        for layer in self.encoders:
            for _ in range(self.num_blocks_share):
                xs, _ = layer(xs, m8, pos_emb)
        xs = ops.layer_norm(xs, self.after_norm.weight, self.after_norm.bias, self.after_norm.eps, sole_consumer=True)
        return xs, masks, pos_emb


class TransformerEncoder(torch.nn.Module):
    def __init__(self, input_size: int, input_layer: str = "conv2d", pos_enc_layer_type: str = "abs_pos",
                 d_model: int = 256, dropout_rate: float = 0.1, attention_heads: int = 4, linear_units: int = 2048,
                 activation_type: str = "swish", macaron_style: bool = True, use_cnn_module: bool = True,
                 cnn_module_kernel: int = 15, causal: bool = False, use_adapter: bool = False, down_size: int = 64,
                 scalar: float = 0.1, num_blocks: int = 6, num_blocks_share: int = 1,
                 global_cmvn: torch.nn.Module = None):
        super().__init__()
        self._output_size = d_model
        self.num_blocks_share = num_blocks_share
        sub = {"linear": LinearNoSubsampling, "conv2d": Conv2dSubsampling4, "conv2d6": Conv2dSubsampling6,
               "conv2d8": Conv2dSubsampling8}
        if input_layer not in sub:
            raise ValueError("unknown input_layer: " + input_layer)
        pos = {"abs_pos": PositionalEncoding, "rel_pos": RelPositionalEncoding}
        if pos_enc_layer_type == "no_pos":
            # the reference selects a class it neither defines nor imports (encoder.py:165-166): same failure, same type
            raise NameError("name 'NoPositionalEncoding' is not defined")
        if pos_enc_layer_type not in pos:
            raise ValueError("unknown pos_enc_layer: " + pos_enc_layer_type)
        self.global_cmvn = global_cmvn
        self.embed = sub[input_layer](input_size, d_model, pos[pos_enc_layer_type](d_model))
        self.encoders = _layers(d_model, dropout_rate, attention_heads, linear_units, activation_type, macaron_style,
                                use_cnn_module, cnn_module_kernel, causal, use_adapter, num_blocks // num_blocks_share,
                                down_size, scalar)
        self.after_norm = torch.nn.LayerNorm(d_model, eps=1e-5)

    def output_size(self) -> int:
        return self._output_size

    def forward(self, xs: torch.Tensor, masks: torch.Tensor):
        """xs (B,T,F) features, masks (B,1,T) bool -> (encoded (B,T',d), masks (B,1,T'), pos_emb (1,T',d))."""
        if self.global_cmvn is not None:
            xs = self.global_cmvn(xs)
        xs, masks, pos_emb = self.embed(xs, masks)
        m8 = ops.mask_bytes(masks)
        hooks = getattr(self, "grad_ready_hooks", None)           # {layer index: callback}, set by TrainEngine (multi-GPU)
        ahead = []
        if ops.POS_PROJ_AHEAD and xs.is_cuda:
            # every layer's linear_pos(pos_emb) on the side stream now (ops.POS_PROJ_AHEAD); a layer waits for its own event
            main, side = torch.cuda.current_stream(), ops.decoder_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for layer in self.encoders:
                    att = getattr(layer, "self_attn", None)
                    if hasattr(att, "linear_pos"):
                        pp = ops.pos_proj(pos_emb, att.linear_pos.weight)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        att._pp_ahead = (pp, ev)
                        ahead.append(att)
        try:
            ops.stamp("fwd: input layer done")
            pre, normed_out = None, False
            for i, layer in enumerate(self.encoders):
                if i % 3 == 0:
                    ops.stamp(f"fwd: encoder layer {i} starts")
                    ops.stamp_grad(xs, f"bwd: encoder layers >= {i} done")
                if hooks and i in hooks and xs.requires_grad:
                    xs.register_hook(lambda g, cb=hooks[i]: cb())     # gradient of layer i's input ready = layers >= i done
                    xs = ops.cut(xs, f"enc{i}")                       # segmented capture: the tape ends here (identity otherwise)
                # a layer's norm_final and the norm that follows it (the next layer's first pre-norm, or after_norm behind the
                # last layer) as one launch: the layer hands back its un-normalised output, the pair is applied here.  Not across
                # a boundary that carries a gradient hook / tape cut (the multi-rank overlap points): those need the tensor between.
                nxt = self.encoders[i + 1] if i + 1 < len(self.encoders) else None
                boundary_hooked = bool(hooks) and (i + 1) in hooks
                fuse = (self.num_blocks_share == 1 and ops.ln_pair_ok(xs) and getattr(layer, "conv_module", None) is not None
                        and not boundary_hooked and (nxt is None or getattr(nxt, "feed_forward_macaron", None) is not None))
                if not fuse:
                    for _ in range(self.num_blocks_share):
                        xs, _ = layer(xs, m8, pos_emb, pre=pre)
                        pre = None
                    continue
                xs, _ = layer(xs, m8, pos_emb, pre=pre, defer_final=True)
                nf = layer.norm_final
                if nxt is None:
                    an = self.after_norm
                    xs = ops.layer_norm_pair(xs, nf.weight, nf.bias, nf.eps, an.weight, an.bias, an.eps, want_first=False, sole_consumer=True)
                    normed_out = True
                else:
                    nm = nxt.norm_ff_macaron
                    # (fuse_fwd: `pre` goes to the next layer's first feed-forward as (residual, input) and nowhere else)
                    pre = ops.layer_norm_pair(xs, nf.weight, nf.bias, nf.eps, nm.weight, nm.bias, nm.eps, want_first=True, sole_consumer=True,
                                              fuse_fwd=True)
                    xs = pre[0]
        finally:
            for att in ahead:
                att._pp_ahead = None
            if ahead:
                torch.cuda.current_stream().wait_stream(ops.decoder_stream())
        if not normed_out:
            xs = ops.layer_norm(xs, self.after_norm.weight, self.after_norm.bias, self.after_norm.eps, sole_consumer=True)
        return xs, masks, pos_emb
