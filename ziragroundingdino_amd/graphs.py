"""hipGraph capture for the launch-bound, gradient-free parts of the step.

The frozen front end of ZiRa (Swin backbone, BERT) is ~2500 small kernels per step whose
launch cost on the host exceeds their run time on an MI355X; nothing in it depends on the host
and no gradient flows into it.  ``GraphedNoGrad`` captures such a callable once per input
signature into a HIP graph (``torch.cuda.CUDAGraph``; random ops such as the backbone's
stochastic depth keep working through the graph-safe Philox generator) and replays it with one
launch.  Inputs are copied into static buffers.  What a capture bakes in besides the shapes is part
of the cache key: the train / eval mode of the wrapped modules (stochastic depth only runs in training)
and the autocast state (an fp32 capture must not be replayed under bf16 autocast, or the reverse).
The outputs handed back are copies of the graph's static output buffers, so features saved for a
later backward (the side branches' weight gradients) survive further forwards -- gradient
accumulation, or an evaluation pass between forward and backward.
"""
import torch


def _flatten(obj, out):
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _flatten(o, out)
    elif isinstance(obj, dict):
        for k in obj:
            _flatten(obj[k], out)
    return out


def _clone_tree(obj):
    if torch.is_tensor(obj):
        return obj.clone()
    if isinstance(obj, (list, tuple)):
        return type(obj)(_clone_tree(o) for o in obj)
    if isinstance(obj, dict):
        return {k: _clone_tree(v) for k, v in obj.items()}
    return obj


class GraphedNoGrad:
    """fn: tensor-in / tensor-out callable; modules: the nn.Modules it runs (their .training flags are
    part of the cache key); clone_outputs=False hands out the static buffers themselves (valid until
    the next replay of the same signature only)."""

    def __init__(self, fn, modules=(), max_signatures=8, warmup=2, clone_outputs=True):
        self.fn = fn
        self.modules = tuple(modules)
        self.max_signatures = max_signatures
        self.warmup = warmup
        self.clone_outputs = clone_outputs
        self._cache = {}

    def _mode_key(self):
        training = tuple(bool(m.training) for m in self.modules)
        autocast = (torch.is_autocast_enabled(), torch.get_autocast_gpu_dtype() if torch.is_autocast_enabled() else None)
        from .transformer import Switches
        return training, autocast, Switches.gemm_arith   # (a capture bakes the arithmetic of the frozen products in)

    def __call__(self, *args):
        tensors = _flatten(args, [])
        if not tensors or not tensors[0].is_cuda or torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            with torch.no_grad():
                return self.fn(*args)
        key = (tuple((tuple(t.shape), t.dtype, t.device.index) for t in tensors), self._mode_key())
        entry = self._cache.get(key)
        if entry is None:
            if len(self._cache) >= self.max_signatures:  # unbounded shape variety: stay eager
                with torch.no_grad():
                    return self.fn(*args)
            entry = self._capture(args, tensors)
            self._cache[key] = entry
        static_in, graph, static_out = entry
        for m in self.modules:      # derived weights (the bf16 planes of the split-bf16 arithmetic) follow their parameters IN PLACE
            refresh = getattr(m, "refresh_derived", None)
            if refresh is not None:
                refresh()
        for s, t in zip(static_in, tensors):
            s.copy_(t, non_blocking=True)
        graph.replay()
        return _clone_tree(static_out) if self.clone_outputs else static_out

    def _capture(self, args, tensors):
        static_in = [t.clone() for t in tensors]
        it = iter(static_in)

        def rebuild(obj):
            if torch.is_tensor(obj):
                return next(it)
            if isinstance(obj, (list, tuple)):
                return type(obj)(rebuild(o) for o in obj)
            if isinstance(obj, dict):
                return {k: rebuild(v) for k, v in obj.items()}
            return obj

        static_args = rebuild(args)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):
                self.fn(*static_args)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            static_out = self.fn(*static_args)
        return static_in, graph, static_out


class _EncoderLayerPiece(torch.nn.Module):
    """Text enhancer + deformable image layer of one encoder layer with a tensor-only signature
    (the fusion block in front of them runs eagerly, see GraphedTransformer).  The level tables
    are constants of the piece (cached device tensors)."""

    def __init__(self, encoder, layer_id, spatial_shapes, level_start_index, no_padding=False):
        super().__init__()
        self.encoder, self.layer_id, self.no_padding = encoder, layer_id, no_padding
        self.spatial_shapes, self.level_start_index = spatial_shapes, level_start_index

    def forward(self, output, memory_text, pos, reference_points, key_padding_mask,
                text_attention_mask, pos_text, text_self_attention_masks):
        return self.encoder.forward_layer(
            self.layer_id, output, memory_text, pos, reference_points, self.spatial_shapes,
            self.level_start_index, None if self.no_padding else key_padding_mask, text_attention_mask,
            pos_text, text_self_attention_masks, fuse=False)


class _SelectDecodePiece(torch.nn.Module):
    """Two-stage query selection + the whole decoder with a tensor-only signature."""

    def __init__(self, transformer, shapes, spatial_shapes, level_start_index, no_padding=False):
        super().__init__()
        self.transformer, self.shapes, self.no_padding = transformer, shapes, no_padding
        self.spatial_shapes, self.level_start_index = spatial_shapes, level_start_index

    def forward(self, memory, memory_text, mask_flatten, lvl_pos, valid_ratios, text_token_mask):
        text_dict = {"encoded_text": memory_text, "text_token_mask": text_token_mask}
        hs, refs, hs_enc, ref_enc, init_box = self.transformer.select_and_decode(
            memory, mask_flatten, lvl_pos, self.shapes, self.spatial_shapes, self.level_start_index,
            valid_ratios, text_dict, no_padding=self.no_padding, sort_for_topk=True)  # topk faults on replay
        self.n_hs, self.n_refs = len(hs), len(refs)
        return (*hs, *refs, hs_enc, ref_enc, init_box)


class _FusionPiece(torch.nn.Module):
    """One image<->text fusion block (BiAttentionBlock) with a tensor-only signature."""

    def __init__(self, block, no_padding=False):
        super().__init__()
        self.block, self.no_padding = block, no_padding

    def forward(self, v, l, mask_v, mask_l):
        return self.block(v=v, l=l, attention_mask_v=None if self.no_padding else mask_v, attention_mask_l=mask_l)


class _DecoderPiece(torch.nn.Module):
    """The decoder alone (query selection stays eager), tensor-only signature."""

    def __init__(self, transformer, spatial_shapes, level_start_index, no_padding=False):
        super().__init__()
        self.transformer, self.no_padding = transformer, no_padding
        self.spatial_shapes, self.level_start_index = spatial_shapes, level_start_index

    def forward(self, tgt, refpoint_embed, memory, memory_text, mask_flatten, lvl_pos, valid_ratios, text_token_mask):
        text_dict = {"encoded_text": memory_text, "text_token_mask": text_token_mask}
        hs, refs = self.transformer.run_decoder(tgt, refpoint_embed, memory, mask_flatten, lvl_pos,
                                                self.spatial_shapes, self.level_start_index, valid_ratios,
                                                text_dict, no_padding=self.no_padding)
        self.n_hs, self.n_refs = len(hs), len(refs)
        return (*hs, *refs)


def _graph(mod, args, shared=()):
    """Graph ``mod`` on copies of ``args``; the tensors listed in ``shared`` become static inputs AS THEY ARE (a replay
    skips the copy of an argument that already is the static input: ``_shared_inputs``)."""
    sample = tuple(x if any(x is y for y in shared) else x.detach().clone().requires_grad_(x.requires_grad) for x in args)
    return torch.cuda.make_graphed_callables(mod, sample, num_warmup_iters=3, allow_unused_input=True)


def _shared_inputs(entry, **tensors):
    """Per-step constants that every piece reads (the level position embedding -- 45 MB at the bench size -- reference
    points, masks): one persistent buffer each, filled once per call and handed to all seven graphs as their static input,
    instead of one 45 MB copy per graph and step.  Only tensors without gradient qualify."""
    out = []
    for name, x in tensors.items():
        assert not x.requires_grad, name
        buf = entry["shared"].get(name)
        if buf is None:
            buf = entry["shared"][name] = torch.empty_like(x)
        buf.copy_(x)
        out.append(buf)
    return out


class GraphedTransformer:
    """Forward AND backward of the (frozen-weight) cross-modal transformer replayed from HIP
    graphs (``torch.cuda.make_graphed_callables``): in eager mode the host needs ~31 ms to
    enqueue the ~1500 small kernels of the forward alone, longer than the GPU needs to run them.

    The transformer is cut into seven pieces that can replay from graphs -- text enhancer + deformable layer of each
    encoder layer, and query selection + decoder -- each with its own pair of graphs and memory
    pool (since round 4 only the decoder piece does by default: see ``graph_encoder`` below); the six image<->text fusion blocks between them stay eager.  Two things fault on the second
    replay on ROCm 7.2 / torch 2.10 and are avoided: ``torch.topk`` inside a graph (the graphed query
    selection takes the first k of a stable descending sort instead: same indices unless logits tie)
    and graphs that contain three or more BiAttention blocks (found by bisection in round 2).
    One set of graphs per input signature (image size / caption length); further signatures
    run eagerly after ``max_signatures``."""

    # Class-level switches: which pieces replay from graphs.  The decoder piece is ~1000 launches of a few microseconds: graphed,
    # 36.6-36.7 against 39.6-39.9 ms per step launched eagerly.  The six encoder pieces are ~120 launches for ~4 ms of GPU time
    # each -- the host stays ahead of them on any machine -- and run 0.5 ms per step FASTER launched eagerly (36.57 / 36.75
    # against 37.13 / 37.27 ms with them graphed; both eager 36.99 / 36.74: scripts/ab_step.py graph_encoder=0|1
    # graph_decoder=0|1, alternating processes on one box).  ROUND 6: with the fused f16x2 FFN and the panel GEMMs an encoder
    # piece is ~1.0 / 1.5 ms of GPU time forward / backward instead of ~4, and the host no longer stays ahead of the eager
    # launches on every box: 26.0-32.9 ms per step launched eagerly (noisy) against 25.3-25.6 replayed on a box with a busy host,
    # equal on others -- the encoder pieces replay from graphs again by default (graph_encoder = False remains supported).
    graph_encoder = True
    graph_decoder = True
    graph_fusion = False     # True: each fusion block replays from its own graph pair as well (300-step soak passes; SLOWER: 37.2-37.4 against 36.5-36.7 ms per step)
    graph_selection = False  # two-stage query selection runs eagerly with torch.topk -- the indices the eager path and the
                             # reference pick, ties included; True: inside the decoder's graph, as the first k of a stable sort

    def __init__(self, transformer, max_signatures=2):
        self.transformer = transformer
        self.max_signatures = max_signatures
        self._cache = {}
        self._eager_keys = set()     # signatures whose capture was refused: they run eagerly from then on
        self._versions = None

    def _refresh_derived_weights(self):
        """A replay re-runs no Python, so buffers DERIVED from parameters (the MSDA modules' concatenated query
        projection) must follow in-place changes of those parameters (``copy_``, ``load_state_dict`` into a live model)
        here: cheap version check per call, refresh in place when anything moved."""
        # (the parameters are listed afresh every call: load_state_dict(assign=True) and module surgery REPLACE parameter
        #  objects, whose versions a cached list would never see; the walk costs ~0.1 ms for the transformer's 400 tensors)
        ver = 0
        for p in self.transformer.parameters():
            ver += p._version + (p.data_ptr() & 0xFFFFF)
        if ver != self._versions:
            if self._versions is not None:
                for m in self.transformer.modules():
                    if hasattr(m, "refresh_fused_projection"):
                        m.refresh_fused_projection()
            self._versions = ver

    def _capture(self, piece, args, shared, key):
        """``_graph`` with a way out: a refused capture (out of memory for the private pools, an op that cannot be
        captured, another thread touching the GPU during a global-mode capture) must not end a task -- the signature
        is marked eager, logged once, and the piece runs uncaptured in this same process."""
        if key in self._eager_keys:
            return piece
        try:
            return _graph(piece, args, shared)
        except Exception as exc:   # noqa: BLE001 -- whatever the capture raised
            if torch.cuda.is_current_stream_capturing():
                raise              # (still inside a capture: nothing sane can run on this stream)
            torch.cuda.synchronize()
            self._eager_keys.add(key)
            print("[GraphedTransformer] hipGraph capture refused (%s: %s); this input signature runs eagerly"
                  % (type(exc).__name__, str(exc).splitlines()[0] if str(exc) else ""), flush=True)
            return piece

    def __call__(self, srcs, masks, poss, text_dict, no_padding=False):
        t = self.transformer
        from .transformer import Switches
        key = (tuple((tuple(x.shape), x.requires_grad) for x in srcs),
               tuple(text_dict["encoded_text"].shape), text_dict["encoded_text"].requires_grad, bool(no_padding),
               torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda"),   # a capture bakes the dtype path in
               Switches.gemm_arith)                                                    # and the arithmetic of the frozen products
        self._refresh_derived_weights()
        if key in self._eager_keys or (key not in self._cache and len(self._cache) >= self.max_signatures):
            hs, refs, hs_enc, ref_enc, init_box, _ = t(srcs, masks, None, poss, None, None, text_dict,
                                                       no_padding=no_padding)
            return hs, refs, hs_enc, ref_enc, init_box

        (src, mask_flat, lvl_pos, shapes, spatial_shapes, level_start_index,
         valid_ratios) = t.prepare_inputs(srcs, masks, poss)
        enc = t.encoder
        memory_text = text_dict["encoded_text"]
        text_attention_mask = ~text_dict["text_token_mask"]
        reference_points, pos_text = enc.prepare(shapes, valid_ratios, memory_text, None,
                                                 text_dict["position_ids"], src.device)
        tsm = text_dict["text_self_attention_masks"]

        entry = self._cache.get(key)
        if entry is None:
            entry = self._cache[key] = {"layers": [], "fusion": [], "decode": None, "shared": {}}
        shared = ()
        consts = (lvl_pos, reference_points, mask_flat, text_attention_mask, pos_text, tsm)
        if all(torch.is_tensor(x) and not x.requires_grad for x in consts):
            shared = _shared_inputs(entry, lvl_pos=lvl_pos, reference_points=reference_points, mask_flat=mask_flat,
                                    text_attention_mask=text_attention_mask, pos_text=pos_text, tsm=tsm)
            lvl_pos, reference_points, mask_flat, text_attention_mask, pos_text, tsm = shared
        output = src
        for i in range(len(enc.layers)):
            if enc.fusion_layers and self.graph_fusion:   # one graph pair per block (developer switch)
                fargs = (output, memory_text, mask_flat, text_attention_mask)
                if len(entry["fusion"]) <= i:
                    entry["fusion"].append(self._capture(_FusionPiece(enc.fusion_layers[i], no_padding), fargs, (), key))
                output, memory_text = entry["fusion"][i](*fargs)
            elif enc.fusion_layers:  # eager: graphs with several BiAttention blocks fault on replay (see class doc)
                output, memory_text = enc.fusion_layers[i](v=output, l=memory_text,
                                                           attention_mask_v=None if no_padding else mask_flat,
                                                           attention_mask_l=text_attention_mask)
            args = (output, memory_text, lvl_pos, reference_points, mask_flat, text_attention_mask,
                    pos_text, tsm)
            if len(entry["layers"]) <= i:
                piece = _EncoderLayerPiece(enc, i, spatial_shapes, level_start_index, no_padding)
                entry["layers"].append(self._capture(piece, args, shared, key) if self.graph_encoder else piece)
            output, memory_text = entry["layers"][i](*args)
        text_dict["encoded_text"] = memory_text
        if entry["decode"] is not None:      # (not while pieces are still being captured: the hook's owner falls back)
            t.fire_after_encoder()

        if not self.graph_selection:   # selection eager, decoder layers graphed
            refpoint_embed, tgt, init_box, hs_enc, ref_enc = t.select_queries(output, mask_flat, shapes, text_dict)
            args = (tgt, refpoint_embed, output, memory_text, mask_flat, lvl_pos, valid_ratios,
                    text_dict["text_token_mask"])
            if entry["decode"] is None:
                piece = _DecoderPiece(t, spatial_shapes, level_start_index, no_padding)
                entry["decode"] = (self._capture(piece, args, shared, key) if self.graph_decoder else piece, piece)
            graphed, piece = entry["decode"]
            out = graphed(*args)
            return list(out[:piece.n_hs]), list(out[piece.n_hs:]), hs_enc, ref_enc, init_box
        args = (output, memory_text, mask_flat, lvl_pos, valid_ratios, text_dict["text_token_mask"])
        if entry["decode"] is None:
            piece = _SelectDecodePiece(t, shapes, spatial_shapes, level_start_index, no_padding)
            entry["decode"] = (self._capture(piece, args, shared, key) if self.graph_decoder else piece, piece)
        graphed, piece = entry["decode"]
        out = graphed(*args)
        nh, nr = piece.n_hs, piece.n_refs
        hs, refs = list(out[:nh]), list(out[nh:nh + nr])
        hs_enc, ref_enc, init_box = out[nh + nr:nh + nr + 3]
        return hs, refs, hs_enc, ref_enc, init_box
