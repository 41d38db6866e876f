"""CTC / attention hybrid ASR model with the reference's public surface
(/root/reference/openeat/models/asr_model.py): constructor kwargs = YAML
``model_conf`` keys, ``forward(features, features_length, targets,
targets_length) -> (loss, acc)``, the four decoding entry points, and the same
flat state-dict keys.  All tensor arithmetic runs in the gfx950 kernels."""
from collections import defaultdict
import os
from typing import List, Optional, Tuple

import torch

from openeat_amd import ops
from openeat_amd import planes as _planes
from openeat_amd.modules.cmvn import GlobalCMVN
from openeat_amd.modules.ctc import CTC
from openeat_amd.modules.decoder import BiTransformerDecoder
from openeat_amd.modules.encoder import TransformerEncoder
from openeat_amd.modules.label_smoothing_loss import LabelSmoothingLoss
from openeat_amd.utils.cmvn import load_cmvn
from openeat_amd.utils.common import (IGNORE_ID, add_sos_eos, log_add, remove_duplicates_and_blank, reverse_pad_list)
from openeat_amd.utils.mask import make_pad_mask, mask_finished_preds, mask_finished_scores, subsequent_mask


# The CTC prefix recursion on the device (beam.hip, one wave per utterance); 0: the native host implementation
# (beam_host.cpp, same algorithm, the device kernel's checker).
DEVICE_BEAM = os.environ.get("OE_DEVICE_BEAM", "1") != "0"
DECODE_GRAPHS = os.environ.get("OE_DECODE_GRAPHS", "0") == "1"
DECODE_GRAPH_SLOTS = 8
ATT_INPUTS_KERNEL = os.environ.get("OE_ATT_INPUTS_KERNEL", "1") != "0"      # decoder token bookkeeping as one launch (fixed widths)


def _graph_call(cache, key, fn, args):
    """fn(*args) through a per-key HIP graph: first call eager (lazy state must exist before a capture) and captured for
    the next time, later calls copy the arguments into the capture's static inputs and replay.  Outputs are the capture's
    own tensors (valid until the next replay of the same key).  A key whose capture failed stays eager."""
    rec = cache.get(key, False)
    if rec is None:
        return fn(*args)
    if rec is False:
        out = fn(*args)                                            # the real result of this call
        static = tuple(a.clone() for a in args)
        g = torch.cuda.CUDAGraph()
        try:
            torch.cuda.synchronize()
            # the capture neither reads pre-split operands that eager code made (the registry's FIFO would free them under the
            # graph: the position table's planes, round 3's red replay) nor leaves its own - unwritten until the first replay - behind
            with _planes.capture_scope(), torch.cuda.graph(g):
                sout = fn(*static)
            cache[key] = (g, static, sout)
        except Exception as e:                                     # noqa: BLE001 - whatever the capture objected to: stay eager
            import warnings
            warnings.warn(f"decode stage {key[0]} {key[1]} is not capturable ({type(e).__name__}: {str(e)[:200]}); it keeps running eagerly")
            cache[key] = None
        stage = [k for k in cache if k[0] == key[0]]
        for k in stage[:-DECODE_GRAPH_SLOTS]:
            del cache[k]
        return out
    g, static, sout = rec
    for s_, a in zip(static, args):
        s_.copy_(a)
    g.replay()
    return sout


class ASRModel(torch.nn.Module):
    def __init__(self, input_size: int, vocab_size: int, is_json_cmvn: bool = True, cmvn_file: str = None,
                 cn_cmvn_file: str = None, en_cmvn_file: str = None, encoder_num_blocks: int = 12,
                 encoder_num_blocks_share: int = 1, decoder_num_blocks: int = 6, r_decoder_num_blocks: int = 0,
                 decoder_num_blocks_share: int = 1, input_layer: str = "conv2d", pos_enc_layer_type: str = "rel_pos",
                 d_model: int = 256, attention_heads: int = 4, linear_units: int = 1024, dropout_rate: float = 0.1,
                 activation_type: str = "swish", macaron_style: bool = True, use_cnn_module: bool = True,
                 cnn_module_kernel: int = 15, causal: bool = False, encoder_use_adapter: bool = False,
                 decoder_use_adapter: bool = False, down_size: int = 64, scalar: float = 0.1, ctc_weight: float = 0.3,
                 lsm_weight: float = 0.1, reverse_weight: float = 0.0, length_normalized_loss: bool = False,
                 ignore_id=IGNORE_ID):
        super().__init__()
        self.input_size = input_size
        self.vocab_size = vocab_size
        self.sos = vocab_size - 1
        self.eos = vocab_size - 1
        self.ignore_id = ignore_id
        self.ctc_weight = ctc_weight
        self.reverse_weight = reverse_weight
        global_cmvn = None
        if cmvn_file is not None:
            mean, istd = load_cmvn(cmvn_file, is_json_cmvn)
            global_cmvn = GlobalCMVN(torch.from_numpy(mean).float(), torch.from_numpy(istd).float())
        self.encoder = TransformerEncoder(
            input_size, input_layer, pos_enc_layer_type, d_model, dropout_rate, attention_heads, linear_units,
            activation_type, macaron_style, use_cnn_module, cnn_module_kernel, causal, encoder_use_adapter, down_size,
            scalar, num_blocks=encoder_num_blocks, num_blocks_share=encoder_num_blocks_share, global_cmvn=global_cmvn)
        self.ctc = CTC(vocab_size, d_model, length_normalized_loss)
        self.decoder = BiTransformerDecoder(
            vocab_size, d_model, dropout_rate, attention_heads, linear_units, decoder_use_adapter, down_size, scalar,
            num_blocks=decoder_num_blocks, r_num_blocks=r_decoder_num_blocks, num_blocks_share=decoder_num_blocks_share)
        self.criterion_att = LabelSmoothingLoss(size=vocab_size, padding_idx=ignore_id, smoothing=lsm_weight,
                                                normalize_length=length_normalized_loss)

    # ------------------------------------------------------------------ train --
    def _encode(self, features, features_length):
        _planes.new_pass()                      # (decode entry points come through here without forward()'s predrop_clear)
        masks = ~make_pad_mask(features_length, features.size(1)).unsqueeze(1)      # (B,1,T)
        return self.encoder(features, masks)

    def forward(self, features: torch.Tensor, features_length: torch.Tensor, targets: torch.Tensor,
                targets_length: torch.Tensor) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """asr_model.py:126-157."""
        assert targets_length.dim() == 1, targets_length.shape
        assert (features.shape[0] == features_length.shape[0] == targets.shape[0] == targets_length.shape[0]), \
            (features.shape, features_length.shape, targets.shape, targets_length.shape)
        ops.predrop_clear()                     # a new tape starts: nothing of the previous backward may be picked up
        par = ops.PARALLEL_DECODERS and features.is_cuda and self.ctc_weight < 1
        prep = None
        if par:
            # the decoders' token bookkeeping (sos/eos, reversal, masks: ~60 tiny launches that depend on the targets only)
            # runs on the decoder stream while the encoder runs here; it is joined where the decoders start
            main, side = torch.cuda.current_stream(), ops.decoder_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                prep = self._att_inputs(targets, targets_length)
        encoder_out, encoder_mask, _ = self._encode(features, features_length)
        if par:
            main.wait_stream(side)
            for t in prep:
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
        ops.stamp("fwd: encoder done")
        ops.stamp_grad(encoder_out, "bwd: heads done")
        hooks = getattr(self, "grad_ready_hooks", None)          # set by TrainEngine for multi-GPU overlap
        if hooks and encoder_out.requires_grad:
            cb = hooks["encoder_out"]
            encoder_out.register_hook(lambda g, cb=cb: cb())      # returns None: the gradient is not modified
            encoder_out = ops.cut(encoder_out, "heads")           # segmented capture: the tape ends here (identity otherwise)
        encoder_out_lens = encoder_mask.squeeze(1).sum(1)
        if par:
            # the CTC head (a few chip-filling launches) on a third stream beside the decoders' latency-bound chains
            main, side = torch.cuda.current_stream(), ops.ctc_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.stamp("fwd: ctc head starts (its stream)")
                loss_ctc = self.ctc(encoder_out, encoder_out_lens, targets, targets_length)
                ops.stamp("fwd: ctc head done (its stream)")
            l_loss, r_loss, acc = self._att_losses(encoder_out, encoder_mask, targets, targets_length, prep)
            main.wait_stream(side)
            loss_ctc.record_stream(main)
            return self._joint_loss(loss_ctc, l_loss, r_loss), acc
        loss_ctc = self.ctc(encoder_out, encoder_out_lens, targets, targets_length)
        if self.ctc_weight < 1:
            l_loss, r_loss, acc = self._att_losses(encoder_out, encoder_mask, targets, targets_length)
            return self._joint_loss(loss_ctc, l_loss, r_loss), acc
        return loss_ctc, None

    def _joint_loss(self, loss_ctc, l_loss, r_loss):
        """asr_model.py:150-157 with :196-198 folded in: ctc_weight * ctc + (1 - ctc_weight) * (l * (1 - rw) + r * rw)."""
        if l_loss.is_cuda and l_loss.dtype == torch.float32:
            return ops.combine_losses(l_loss, r_loss, loss_ctc, self.ctc_weight, self.reverse_weight)
        loss_att = l_loss if r_loss is None else l_loss * (1 - self.reverse_weight) + r_loss * self.reverse_weight
        return self.ctc_weight * loss_ctc + (1 - self.ctc_weight) * loss_att

    def _att_inputs(self, ys_pad, ys_pad_lens):
        """asr_model.py:162-176: decoder inputs / targets of both directions and the target mask (token bookkeeping only)."""
        from openeat_amd.utils import common
        if ATT_INPUTS_KERNEL and common.STATIC_SHAPES and ys_pad.is_cuda and ys_pad.dtype == torch.int32 and ys_pad_lens.dtype == torch.int32 \
                and ys_pad.dim() == 2 and ys_pad.shape[1] <= 8000:
            # fixed width (a captured step): one launch instead of ~70 index-arithmetic launches
            from openeat_amd import hip
            B, L = ys_pad.shape
            W = L + 1
            rev = self.reverse_weight > 0
            out = torch.empty(4 if rev else 2, B, W, dtype=torch.long, device=ys_pad.device)
            tgt_mask = torch.empty(B, W, W, dtype=torch.bool, device=ys_pad.device)
            hip.call("oe_att_inputs", ys_pad.contiguous(), ys_pad_lens.contiguous(), B, L, self.sos, self.eos, self.ignore_id, out[0], out[1],
                     out[2] if rev else None, out[3] if rev else None, tgt_mask)
            return out[0], out[1], tgt_mask, (out[2] if rev else None), (out[3] if rev else None)
        ys_in_pad, ys_out_pad = add_sos_eos(ys_pad, self.sos, self.eos, self.ignore_id)
        ys_in_lens = ys_pad_lens + 1
        L = ys_in_pad.size(1)
        tgt_mask = (~make_pad_mask(ys_in_lens, L)).unsqueeze(1) & subsequent_mask(L, device=ys_in_pad.device).unsqueeze(0)
        r_ys_in_pad = r_ys_out_pad = None
        if self.reverse_weight > 0:
            r_ys_pad = reverse_pad_list(ys_pad, ys_pad_lens, float(self.ignore_id))
            r_ys_in_pad, r_ys_out_pad = add_sos_eos(r_ys_pad, self.sos, self.eos, self.ignore_id)
        return ys_in_pad, ys_out_pad, tgt_mask, r_ys_in_pad, r_ys_out_pad

    def _calc_att_loss(self, encoder_out, encoder_mask, ys_pad, ys_pad_lens, prep=None):
        """asr_model.py:159-203; the output layers are fused with the loss."""
        l_loss, r_loss, acc = self._att_losses(encoder_out, encoder_mask, ys_pad, ys_pad_lens, prep)
        if r_loss is None:
            return l_loss, acc
        if l_loss.is_cuda and l_loss.dtype == torch.float32:
            return ops.combine_losses(l_loss, r_loss, None, 0.0, self.reverse_weight), acc
        return l_loss * (1 - self.reverse_weight) + r_loss * self.reverse_weight, acc

    def _att_losses(self, encoder_out, encoder_mask, ys_pad, ys_pad_lens, prep=None):
        """The two decoders' losses of asr_model.py:159-203 before they are mixed: (left, right or None, accuracy)."""
        ys_in_pad, ys_out_pad, tgt_mask, r_ys_in_pad, r_ys_out_pad = prep if prep is not None else self._att_inputs(ys_pad, ys_pad_lens)
        dec = self.decoder

        def left():
            l_hid = dec.left_decoder.hidden(ys_in_pad, tgt_mask, encoder_out, encoder_mask)
            lo = dec.left_decoder.output_layer
            return self.criterion_att.fused_head(l_hid, lo.weight, lo.bias, ys_out_pad)

        if not self.reverse_weight > 0:
            loss_att, n_ok, n_valid = left()
            return loss_att, None, torch.true_divide(n_ok, n_valid)

        def right():
            r_hid = dec.right_decoder.hidden(r_ys_in_pad, tgt_mask, encoder_out, encoder_mask)
            ro = dec.right_decoder.output_layer
            return self.criterion_att.fused_head(r_hid, ro.weight, ro.bias, r_ys_out_pad)[0]

        if ops.PARALLEL_DECODERS and encoder_out.is_cuda:
            # the right decoder on its own stream beside the left one (ops.PARALLEL_DECODERS): fork here, join below
            main, side = torch.cuda.current_stream(), ops.decoder_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.stamp("fwd: right decoder starts (its stream)")
                r_loss = right()
                ops.stamp("fwd: right decoder done (its stream)")
            ops.stamp("fwd: left decoder starts")
            loss_att, n_ok, n_valid = left()
            ops.stamp("fwd: left decoder done")
            main.wait_stream(side)
            r_loss.record_stream(main)
        else:
            loss_att, n_ok, n_valid = left()
            r_loss = right()
        return loss_att, r_loss, torch.true_divide(n_ok, n_valid)

    # ----------------------------------------------------------------- decode --
    def ctc_greedy_search(self, features: torch.Tensor, features_length: torch.Tensor) -> List[List[int]]:
        """asr_model.py:297-326: argmax / eos-fill / collapse all on device; one D2H copy of the result."""
        assert features.shape[0] == features_length.shape[0]
        encoder_out, encoder_mask, _ = self._encode(features, features_length)
        lens = encoder_mask.squeeze(1).sum(1)
        logits = self.ctc.logits(encoder_out)
        B, T, V = logits.shape
        toks, n = ops.ctc_greedy(logits, V, B, T, V, lens, self.eos)
        toks, n = toks.cpu(), n.cpu()
        return [toks[b, : int(n[b])].tolist() for b in range(B)]

    def _ctc_prefix_beam_search(self, features, features_length, beam_size: int):
        """asr_model.py:328-396 (batch of one): log-probs and per-frame top-k on the device, the prefix
        recursion in native host code (oe_ctc_prefix_beam_host: same ordering and float64 arithmetic)."""
        assert features.shape[0] == features_length.shape[0]
        assert features.shape[0] == 1
        encoder_out, _, _ = self._encode(features, features_length)
        top_p, top_i = ops.topk_rows(self.ctc.logits(encoder_out).squeeze(0), beam_size, log_softmax=True)
        from openeat_amd import hip
        if DEVICE_BEAM and beam_size <= 16:
            return hip.ctc_prefix_beam_device(top_p.unsqueeze(0), top_i.unsqueeze(0), None, beam_size)[0], encoder_out
        return hip.ctc_prefix_beam_host(top_p.cpu(), top_i.cpu(), beam_size), encoder_out

    def ctc_prefix_beam_search(self, features, features_length, beam_size: int) -> List[int]:
        hyps, _ = self._ctc_prefix_beam_search(features, features_length, beam_size)
        return hyps[0][0]

    def attention_rescoring(self, features, features_length, beam_size: int, ctc_weight: float = 0.0,
                            reverse_weight: float = 0.0, lm: Optional[torch.nn.Module] = None, lm_weight: float = 0,
                            autoregressive: bool = True, token2char: dict = {}):
        """asr_model.py:418-534."""
        assert features.shape[0] == features_length.shape[0] == 1
        device = features.device
        hyps, encoder_out = self._ctc_prefix_beam_search(features, features_length, beam_size)
        assert len(hyps) == beam_size
        lens = torch.tensor([len(h[0]) for h in hyps], device=device, dtype=torch.long)
        Lm = int(lens.max())
        ori = torch.full((beam_size, Lm), self.ignore_id, dtype=torch.long, device=device)
        for i, h in enumerate(hyps):
            ori[i, : len(h[0])] = torch.tensor(h[0], dtype=torch.long, device=device)
        hyps_pad, _ = add_sos_eos(ori, self.sos, self.eos, self.ignore_id)
        in_lens = lens + 1
        L = hyps_pad.size(1)
        hyps_mask = (~make_pad_mask(in_lens, L)).unsqueeze(1) & subsequent_mask(L, device=device).unsqueeze(0)
        enc = encoder_out.repeat(beam_size, 1, 1)
        enc_mask = torch.ones(beam_size, 1, enc.size(1), dtype=torch.bool, device=device)
        r_pad = reverse_pad_list(ori, lens, self.ignore_id)
        r_hyps_pad, _ = add_sos_eos(r_pad, self.sos, self.eos, self.ignore_id)
        if reverse_weight > 0 and self.decoder.r_num_blocks == 0:
            raise IndexError("reverse_weight > 0 needs r_decoder_num_blocks > 0 (as in the reference)")
        decoder_out, r_decoder_out, pre = self.decoder(enc, enc_mask, hyps_pad, r_hyps_pad, hyps_mask)
        l_lp = ops.log_softmax_rows(decoder_out).cpu().numpy()
        r_lp = ops.log_softmax_rows(r_decoder_out).cpu().numpy() if self.decoder.r_num_blocks > 0 else None
        use_nn_lm = lm_weight > 0 and isinstance(lm, torch.nn.Module)
        lm_lp = None
        if use_nn_lm:
            # asr_model.py:490-499 (the reference calls `lm.encoder(tokens, lengths)`, meaning the LM's forward up to the
            # projection; see openeat_amd/models/language_model.py).  Autoregressive LM: scored on sos + hypothesis.
            if autoregressive:
                lm_lp = lm.log_probs(hyps_pad, in_lens).cpu().numpy()
            else:
                lm_in = ori.masked_fill(ori == self.ignore_id, self.eos)
                lm_lp = lm.log_probs(lm_in, in_lens - 1).cpu().numpy()
        best, best_i = -float("inf"), 0
        for i, (hyp, ctc_score) in enumerate(hyps):
            score = sum(l_lp[i][j][w] for j, w in enumerate(hyp)) + l_lp[i][len(hyp)][self.eos]
            lm_score = 0.0
            if use_nn_lm:
                lm_score = sum(lm_lp[i][j][w] for j, w in enumerate(hyp))
            elif lm_weight > 0 and lm is not None:
                lm_score = lm.score(" ".join(token2char[w] for w in hyp), bos=True, eos=True)
            if reverse_weight > 0:
                r = sum(r_lp[i][len(hyp) - j - 1][w] for j, w in enumerate(hyp)) + r_lp[i][len(hyp)][self.eos]
                score = score * (1 - reverse_weight) + r * reverse_weight
            score += ctc_score * ctc_weight + lm_score * lm_weight
            if score > best:
                best, best_i = score, i
        return hyps[best_i][0], enc, pre

    @torch.no_grad()
    def attention_rescoring_batch(self, features: torch.Tensor, features_length: torch.Tensor, beam_size: int,
                                  ctc_weight: float = 0.0, reverse_weight: float = 0.0,
                                  lm: Optional[torch.nn.Module] = None, lm_weight: float = 0.0,
                                  use_graphs: Optional[bool] = None) -> List[List[int]]:
        """Batched form of attention_rescoring (the reference handles one utterance per call,
        asr_model.py:444): ONE encoder pass and ONE fused log-softmax top-k for the whole batch, the prefix
        recursion of every utterance on its own valid frames (one wave each on the device; OE_DEVICE_BEAM=0: native host
        code), then ONE bi-decoder pass over all B x beam hypotheses and the same score mix (asr_model.py:504-528).
        use_graphs (default: OE_DECODE_GRAPHS, off): replay the two stages from HIP graphs cached per shape."""
        from openeat_amd import hip
        device = features.device
        B = features.shape[0]
        R = B * beam_size
        if reverse_weight > 0 and self.decoder.r_num_blocks == 0:
            raise IndexError("reverse_weight > 0 needs r_decoder_num_blocks > 0 (as in the reference)")
        on_device = DEVICE_BEAM and beam_size <= 16
        graphs = DECODE_GRAPHS if use_graphs is None else bool(use_graphs)
        if on_device and graphs:
            return self._rescoring_batch_graphs(features, features_length, beam_size, ctc_weight, reverse_weight, lm, lm_weight)
        if on_device:
            encoder_out, encoder_mask, pre, plen, ctc_scores, bad = self._rescore_stage1(features, features_length, beam_size)
            Lm = max(int(plen.max()), 1)                           # the one host sync of the n-best stage
            if int(bad):
                raise RuntimeError("oe_ctc_prefix_beam: a prefix exceeded max_len")
            toks, n, mean_len = self._rescore_stage2(encoder_out, encoder_mask, pre, plen, ctc_scores, Lm, beam_size, ctc_weight,
                                                     reverse_weight, lm, lm_weight)
            self.last_nbest_mean_len = float(mean_len)
            toks, n = toks.cpu(), n.cpu().tolist()
            return [toks[b, : n[b]].tolist() for b in range(B)]
        encoder_out, encoder_mask, _ = self._encode(features, features_length)
        lens = encoder_mask.squeeze(1).sum(1)
        top_p, top_i = ops.topk_rows(self.ctc.logits(encoder_out), beam_size, log_softmax=True)
        nbest = hip.ctc_prefix_beam_host_batch(top_p.cpu(), top_i.cpu(), lens.cpu().tolist(), beam_size)
        for b in range(B):                                         # a very short utterance can yield fewer than `beam` prefixes
            while len(nbest[b]) < beam_size:
                nbest[b].append((nbest[b][-1][0], -float("inf")))
        flat = [h for nb in nbest for h in nb]
        self.last_nbest_mean_len = sum(len(h[0]) for h in flat) / max(len(flat), 1)
        hl = torch.tensor([len(h[0]) for h in flat], dtype=torch.long)
        Lm = max(int(hl.max()), 1)
        ori = torch.full((R, Lm), self.ignore_id, dtype=torch.long)
        for i, h in enumerate(flat):
            if h[0]:
                ori[i, : len(h[0])] = torch.tensor(h[0], dtype=torch.long)
        ctc_scores = torch.tensor([h[1] for h in flat], dtype=torch.float64, device=device)
        best = self._rescore_scores(encoder_out, encoder_mask, ori.to(device), hl.to(device), ctc_scores, torch.isinf(ctc_scores), Lm,
                                    beam_size, ctc_weight, reverse_weight, lm, lm_weight).cpu().tolist()
        return [list(nbest[b][best[b]][0]) for b in range(B)]

    # ---- the pieces of the batched rescoring (each of fixed shape given (B, T) resp. (B, T', L): capturable) --------------
    def _rescore_stage1(self, features, features_length, beam_size):
        """Encoder, CTC projection, fused log-softmax top-k, prefix recursion - all on the device, no host sync.
        Returns encoder_out, encoder_mask, prefixes (R, T') int32, lengths (R) int32 (-1: the slot does not exist), CTC
        scores (R) float64, status word."""
        from openeat_amd import hip
        encoder_out, encoder_mask, _ = self._encode(features, features_length)
        lens = encoder_mask.squeeze(1).sum(1)
        top_p, top_i = ops.topk_rows(self.ctc.logits(encoder_out), beam_size, log_softmax=True)
        pre, plen, ctc_scores, bad = hip.ctc_prefix_beam_device(top_p, top_i, lens.to(torch.int32), beam_size, raw=True)
        R = pre.shape[0] * beam_size
        return encoder_out, encoder_mask, pre.view(R, -1), plen.view(R), ctc_scores.view(R), bad

    def _rescore_stage2(self, encoder_out, encoder_mask, pre, plen, ctc_scores, Lm, beam_size, ctc_weight, reverse_weight, lm, lm_weight):
        """The n-best lists as device tensors -> (tokens (B, Lm) of the rescored pick, their lengths (B), mean n-best length).
        Lm >= the longest hypothesis (any padding is ignore_id and masked)."""
        device = pre.device
        B = encoder_out.shape[0]
        missing = plen < 0
        hl = plen.clamp(min=0).long()
        ori = pre[:, :Lm].long()
        ori = ori.masked_fill(torch.arange(Lm, device=device).unsqueeze(0) >= hl.unsqueeze(1), self.ignore_id)
        ctc_scores = ctc_scores.masked_fill(missing, -float("inf"))
        best = self._rescore_scores(encoder_out, encoder_mask, ori, hl, ctc_scores, missing, Lm, beam_size, ctc_weight, reverse_weight,
                                    lm, lm_weight)
        pick = best + torch.arange(B, device=device) * beam_size
        return ori.index_select(0, pick), hl.index_select(0, pick), hl.float().mean()

    def _rescore_scores(self, encoder_out, encoder_mask, ori, hl, ctc_scores, missing, Lm, beam_size, ctc_weight, reverse_weight, lm,
                        lm_weight):
        """asr_model.py:504-528 for all B x beam hypotheses at once: index of the best hypothesis per utterance."""
        device = ori.device
        B = encoder_out.shape[0]
        R = B * beam_size
        hyps_pad, _ = add_sos_eos(ori, self.sos, self.eos, self.ignore_id)
        L = hyps_pad.size(1)
        hyps_mask = (~make_pad_mask(hl + 1, L)).unsqueeze(1) & subsequent_mask(L, device=device).unsqueeze(0)
        enc = encoder_out.repeat_interleave(beam_size, dim=0)
        enc_mask = encoder_mask.repeat_interleave(beam_size, dim=0)
        r_ori = reverse_pad_list(ori, hl, self.ignore_id)
        r_hyps_pad, _ = add_sos_eos(r_ori, self.sos, self.eos, self.ignore_id)
        with ops.causal_self_attention():                                        # hyps_mask is causal: key blocks above the diagonal are skipped
            l_x, r_x, _ = self.decoder(enc, enc_mask, hyps_pad, r_hyps_pad, hyps_mask)
        pos = torch.arange(L, device=device).unsqueeze(0)
        valid = pos < hl.unsqueeze(1)                                            # token positions j < len
        tok = torch.cat([ori, ori.new_full((R, L - Lm), self.ignore_id)], 1).clamp(min=0)

        def seq_score(logits, tokens_at):
            # the token's and <eos>'s log-probability at every position, straight from the logits (no (R, L, V) log-softmax)
            tok_lp, eos_all = ops.logprob_gather(logits, tokens_at, also=self.eos)
            eos_lp = eos_all.gather(1, hl.unsqueeze(1)).squeeze(1)
            return (tok_lp * valid).sum(1).double() + eos_lp.double()

        score = seq_score(l_x, tok)
        if reverse_weight > 0:
            r_tok = torch.cat([r_ori.long(), ori.new_full((R, L - Lm), self.ignore_id)], 1).clamp(min=0)
            score = score * (1 - reverse_weight) + seq_score(r_x, r_tok) * reverse_weight
        score = score + ctc_scores * ctc_weight
        if lm is not None and lm_weight > 0:                                      # neural-LM shallow fusion (asr_model.py:490-527)
            lm_tok = ops.logprob_gather(lm.logits(hyps_pad, hl + 1), tok) if hasattr(lm, "logits") else \
                lm.log_probs(hyps_pad, hl + 1).gather(2, tok.unsqueeze(2)).squeeze(2)
            score = score + (lm_tok * valid).sum(1).double() * lm_weight
        score = score.masked_fill(missing, -float("inf"))         # (0 * -inf above would be nan: the slot is out whatever the weights)
        return score.view(B, beam_size).argmax(1)

    def _rescoring_batch_graphs(self, features, features_length, beam_size, ctc_weight, reverse_weight, lm, lm_weight):
        """The same two stages replayed from HIP graphs (decode is launch-bound: ~1500 small launches for 64 utterances):
        stage 1 keyed by the feature shape, stage 2 by the n-best length rounded up to a multiple of 16; between them the one
        host read of the longest hypothesis.  A shape is run eagerly the first time it is seen and captured for the next;
        a stage that cannot be captured keeps running eagerly.  At most DECODE_GRAPH_SLOTS graphs per stage are kept."""
        from openeat_amd.utils import common
        B = features.shape[0]
        cache = self.__dict__.setdefault("_decode_graphs", {})
        static_before = common.STATIC_SHAPES
        common.STATIC_SHAPES = True                                # label bookkeeping of fixed width: nothing reads a length on the host
        try:
            k1 = ("s1", tuple(features.shape), beam_size)
            out1 = _graph_call(cache, k1, lambda f, fl: self._rescore_stage1(f, fl, beam_size), (features, features_length))
            encoder_out, encoder_mask, pre, plen, ctc_scores, bad = out1
            Lm = max(int(plen.max()), 1)
            if int(bad):
                raise RuntimeError("oe_ctc_prefix_beam: a prefix exceeded max_len")
            Lb = min(-(-Lm // 16) * 16, pre.shape[1])
            k2 = ("s2", tuple(encoder_out.shape), beam_size, Lb, float(ctc_weight), float(reverse_weight), id(lm), float(lm_weight))
            toks, n, mean_len = _graph_call(
                cache, k2, lambda eo, em, p, pl, cs: self._rescore_stage2(eo, em, p, pl, cs, Lb, beam_size, ctc_weight, reverse_weight,
                                                                         lm, lm_weight), (encoder_out, encoder_mask, pre, plen, ctc_scores))
        finally:
            common.STATIC_SHAPES = static_before
        self.last_nbest_mean_len = float(mean_len)
        toks, n = toks.cpu(), n.cpu().tolist()
        return [toks[b, : n[b]].tolist() for b in range(B)]

    def recognize(self, features: torch.Tensor, features_length: torch.Tensor, beam_size: int = 10) -> torch.Tensor:
        """asr_model.py:205-295: batched attention beam search (incl. the reference's un-reordered cache)."""
        assert features.shape[0] == features_length.shape[0]
        device = features.device
        B = features.shape[0]
        encoder_out, encoder_mask, _ = self._encode(features, features_length)
        maxlen, d = encoder_out.size(1), encoder_out.size(2)
        R = B * beam_size
        encoder_out = encoder_out.unsqueeze(1).repeat(1, beam_size, 1, 1).view(R, maxlen, d)
        encoder_mask = encoder_mask.unsqueeze(1).repeat(1, beam_size, 1, 1).view(R, 1, maxlen)
        hyps = torch.full((R, 1), self.sos, dtype=torch.long, device=device)
        scores = torch.tensor([0.0] + [-float("inf")] * (beam_size - 1), device=device).repeat(B).unsqueeze(1)
        end_flag = torch.zeros_like(scores, dtype=torch.bool)
        cache = None
        for i in range(1, maxlen + 1):
            if end_flag.sum() == R:
                break
            hyps_mask = subsequent_mask(i, device=device).unsqueeze(0).repeat(R, 1, 1)
            p, cache, _ = self.decoder.forward_one_step(hyps, hyps_mask, encoder_out, encoder_mask, cache=cache)
            top_k_logp, top_k_index = ops.topk_rows(p, beam_size, log_softmax=True)
            top_k_logp = mask_finished_scores(top_k_logp, end_flag)
            top_k_index = mask_finished_preds(top_k_index, end_flag, self.eos)
            scores = (scores + top_k_logp).view(B, beam_size * beam_size)
            scores, offset_k_index = ops.topk_rows(scores, beam_size)
            scores = scores.view(-1, 1)
            base = torch.arange(B, device=device).view(-1, 1).repeat(1, beam_size) * beam_size * beam_size
            best_k_index = base.view(-1) + offset_k_index.view(-1)
            best_k_pred = torch.index_select(top_k_index.view(-1), dim=-1, index=best_k_index)
            best_hyps_index = best_k_index // beam_size
            hyps = torch.cat((torch.index_select(hyps, 0, best_hyps_index), best_k_pred.view(-1, 1)), dim=1)
            end_flag = torch.eq(hyps[:, -1], self.eos).view(-1, 1)
        scores = scores.view(B, beam_size)
        best_index = torch.argmax(scores, dim=-1).long() + torch.arange(B, dtype=torch.long, device=device) * beam_size
        return torch.index_select(hyps, 0, best_index)[:, 1:]
