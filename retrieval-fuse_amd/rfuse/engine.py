"""Online retrieve -> attend -> refine for batches of 64^3 chunks on one MI355X (one process per GPU).

This is the build's own runner for the data flow of ``RefinementTrainingModule.forward_full``
(reference trainer/train_refinement.py:108-116) + ``network_pred_to_df`` (:242-243), with the reference's OFFLINE
retrieval (util/retrieval.py ``--mode map`` / ``--mode compose``, :222-248) done online on the device:

   raw input chunk  --rf_query_windows-->  query windows (or the padded chunk: conv encoders run fully convolutionally, ops.embed_windows)
                    --fenc_input + normalise-->  unit embeddings                                              (A11, A16)
                    --rf_l2_topk (+ all-gather of the queries, all-to-all of the keys, rf_topk_merge when the DB is sharded)-->  top-2K             (A12)
                    --rf_demote_same_scene-->  top-K (scene, box)                                             (A13)
                    --rf_gather_patches-->  K*64 retrieved 16^3 patches per chunk, normalised, already in the
                                           Unfold3D(16,1) row layout                                          (A15, A16, A4)
   retrieval_backbone(patches) ; unet_backbone(normalised input)                                              (A3, A1)
   patched attention over the patch-major features (no Fold3D materialised)                                   (A5-A7)
   decoder -> tanh -> df = (pred+1) * trunc/2                                                                 (A9, A10)

Everything numeric runs in librfuse_hip.so; this file only sequences launches and owns tensors.
"""
import contextlib
import functools
import io

import torch

import model as rf_model
from . import ops
from .configs import truncations


def _on_engine_device(method):
    """Run a stage with the engine's device current (streams, workspaces and launches all follow the current device), so
    ``RefinementEngine(cfg, 'cuda:1', db)`` works whatever the caller's current device is."""
    @functools.wraps(method)
    def scoped(self, *args, **kwargs):
        if self.device.index == torch.cuda.current_device():
            return method(self, *args, **kwargs)
        with torch.cuda.device(self.device):
            return method(self, *args, **kwargs)
    return scoped


class RefinementEngine:
    def __init__(self, config, device='cuda:0', database=None, quiet=True):
        self.config = config
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.K = config['K']
        self.input_trunc, self.target_trunc = truncations(config)
        sink = io.StringIO() if quiet else None
        with (contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext()):
            self.unet_backbone = rf_model.get_unet_backbone(config)
            self.decoder = rf_model.get_decoder(config)
            self.retrieval_backbone = rf_model.get_retrieval_backbone(config)
            self.patched_attention_block = rf_model.get_attention_block(config)
            self.fenc_input, self.fenc_target = rf_model.get_retrieval_networks(config['retrieval_model'])
        for m in list(self.modules().values()) + ([self.fenc_target] if self.fenc_target is not None else []):
            m.to(self.device).eval()
        self.database = database
        self._side_streams = {}          # one helper stream per caller stream (several batches may be in flight)
        self.serial = False              # True: keep the U-Net backbone on the caller's stream (per-kernel timing wants no overlap)
        self.front_at = 'start'          # refine_stream: where the next batch's front end is issued -- at the 'start' of this batch's back end, behind its encoder launches ('decoders'), or 'unet_decoders': front end at the start, but this batch's chunk-level U-Net forked behind the encoders

    def modules(self):
        return {'unet_backbone': self.unet_backbone, 'decoder': self.decoder, 'retrieval_backbone': self.retrieval_backbone,
                'patched_attention_block': self.patched_attention_block, 'fenc_input': self.fenc_input}

    def load_state_dicts(self, sds):
        for name, sd in sds.items():
            self.modules()[name].load_state_dict(sd)

    def load_checkpoints(self, refinement_ckpt=None, retrieval_ckpt=None, trust_pickle=False):
        """The reference's checkpoint hand-over (trainer/train_refinement.py:295-306, util/retrieval.py:224-225 via util/misc.py:23-36): Lightning-shaped
        ``{'state_dict': {'unet_backbone.network.0...': ..., 'decoder...': ..., 'retrieval_backbone...': ..., 'patched_attention_block...': ...}}`` for the
        refinement networks, ``{'state_dict': {'fenc_input.layers.0...': ..., 'fenc_target...': ...}}`` for the patch encoders (``fenc_target`` embeds
        database patches: ``PatchDatabase.build(config, engine.fenc_target, ...)``).  Paths or loaded checkpoints; strict: a missing or unexpected key
        raises.  Files are read with torch.load(weights_only=True); ``trust_pickle=True`` falls back to full unpickling for checkpoints that carry other Python
        objects -- that EXECUTES code stored in the file, so only for checkpoints of known origin.  -> the attribute names loaded."""
        from . import checkpoint
        loaded = []
        if refinement_ckpt is not None:
            loaded += checkpoint.load_prefixed(self.modules(), refinement_ckpt, checkpoint.REFINEMENT_PREFIXES, self.device, trust_pickle)
        if retrieval_ckpt is not None:
            loaded += checkpoint.load_prefixed({'fenc_input': self.fenc_input, 'fenc_target': self.fenc_target}, retrieval_ckpt,
                                               checkpoint.RETRIEVAL_PREFIXES, self.device, trust_pickle)
        return loaded

    # ---------------------------------------------------------------------------------------------- stages
    @_on_engine_device
    @torch.no_grad()
    def embed_queries(self, input_raw):
        """input_raw [B,S,S,S] un-normalised -> unit query embeddings [B*P, latent] (P = 64 windows per chunk)."""
        g, d = self.config['query_geometry'], self.config['dataset_train']
        pad = 0.0 if self.config['task'] == 'surface_reconstruction' else self.input_trunc
        z = ops.embed_windows(self.fenc_input, input_raw, g['patch_size_input'], g['patch_context_input'], pad, d['input_mean'], d['input_std'])
        return ops.l2_normalize_rows_(z.reshape(z.shape[0], z.shape[1]))

    @_on_engine_device
    @torch.no_grad()
    def retrieve(self, input_raw, query_scene=None, patch_mask=None, q=None):
        """-> (patches [(B*K*64),1,16,16,16] normalised, meta [B*64,K,7]).
        ``patch_mask`` [B,64] bool: the dataset's query-side occupancy filter (reference dataset/patched_scene_dataset.py:28-32).
        Patches it drops (False) are never looked up and keep the truncation value in all K retrieved volumes
        (util/retrieval.py:148,151).  ``q``: the query embeddings if the caller already has them."""
        d = self.config['dataset_train']
        if q is None:
            q = self.embed_queries(input_raw)
        keep = patch_mask.reshape(-1).contiguous() if patch_mask is not None else None
        meta, _, _ = self.database.retrieve(q, self.K, query_scene, keep)
        patches = ops.gather_patches(self.database.volumes, meta, input_raw.shape[0], self.K, self.target_trunc, 1.0,
                                     d['target_mean'], d['target_std'], layout=1)
        return patches, meta

    @_on_engine_device
    @torch.no_grad()
    def normalise_input(self, input_raw):
        """(x - mean) / std of the whole chunk = the window kernel with one window and no context."""
        d = self.config['dataset_train']
        s = input_raw.shape[-1]
        return ops.query_windows(input_raw, s, 0, 0.0, d['input_mean'], d['input_std']).reshape(input_raw.shape[0], 1, s, s, s)

    def _fork_backbone(self, x_in):
        """Launch the U-Net backbone on a second HIP stream: it depends only on the input chunk, is made of small
        launches (1^3..32^3 volumes of a few chunks) that cannot fill 256 CUs, and so runs in the shadow of the
        retrieval path (top-k scan, patch gather, retrieval backbone) on the main stream."""
        main = torch.cuda.current_stream(self.device)
        if self.serial:
            return self.unet_backbone(x_in), main
        side = self._side_streams.get(main.cuda_stream)
        if side is None:
            side = self._side_streams[main.cuda_stream] = torch.cuda.Stream(self.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            x_back = self.unet_backbone(x_in)
        if not torch.cuda.is_current_stream_capturing():         # (a captured graph keeps its private pool alive by itself)
            x_in.record_stream(side)
            x_back.record_stream(main)
        return x_back, side

    @_on_engine_device
    @torch.no_grad()
    def refine_from_patches(self, x_in, patches, gumbel_noise=None, stages=None):
        """x_in [B,1,S,S,S] normalised; patches [(B*K*64),1,16^3] normalised -> df [B,1,64,64,64]."""
        x_back, side = self._fork_backbone(x_in)
        feats = self.retrieval_backbone(patches)                                  # [(B*K*64), nf, 8,8,8], patch-major
        torch.cuda.current_stream(self.device).wait_stream(side)
        return self._attend_and_decode(x_back, feats, gumbel_noise, stages)

    def _attend_and_decode(self, x_back, feats, gumbel_noise, stages=None):
        x = self.patched_attention_block.forward_patch_major(x_back, feats, feats.shape[-1], gumbel_noise)
        df = self.decoder.forward_df(x, self.target_trunc)
        if stages is not None:
            stages.update(x_back=x_back, retrieval_features=feats, x_attn=x, df=df)
        return df

    @_on_engine_device
    @torch.no_grad()
    def refine(self, input_raw, query_scene=None, gumbel_noise=None, use_feature_cache=False, patch_mask=None):
        """The whole online path for a batch of chunks: raw low-res input [B,S,S,S] -> refined TSDF [B,1,64,64,64].

        ``use_feature_cache=True`` (needs ``database.build_feature_cache``) fetches the retrieval-backbone features of the
        retrieved database rows from HBM instead of recomputing them -- an optional serving mode that skips 87 % of the
        FLOPs; results agree with the full path to GroupNorm-statistics rounding."""
        # the backbone is forked first and runs beside everything else of the step (round 2 ran the large-window patch encoders before the
        # fork to keep their F16 MFMAs away from fp32 kernels of the other stream; the cause of that hazard is fixed, DESIGN 4.7, and the order
        # makes no measurable difference: C5 23.9 vs 23.8 ms per step)
        x_back, side = self._fork_backbone(self.normalise_input(input_raw))
        q = self.embed_queries(input_raw)
        if use_feature_cache:
            if self.database.feature_cache is None:
                raise RuntimeError('use_feature_cache=True needs database.build_feature_cache(retrieval_backbone, config) first')
            b = input_raw.shape[0]
            keep = patch_mask.reshape(-1).contiguous() if patch_mask is not None else None
            _, _, idx = self.database.retrieve(q, self.K, query_scene, keep)                # [B*64, K] database row ids; -1 (none) gathers the sentinel row
            order = idx.reshape(b, 64, self.K).permute(0, 2, 1).reshape(-1).contiguous()     # (b, k, slot): the patch-major order
            feats = ops.gather_rows(self.database.feature_cache, order)
        else:
            patches, _ = self.retrieve(input_raw, query_scene, patch_mask, q=q)
            feats = self.retrieval_backbone(patches)
        torch.cuda.current_stream(self.device).wait_stream(side)
        df = self._attend_and_decode(x_back, feats, gumbel_noise)
        self._check_database()
        return df

    def _check_database(self):
        """Sharded database: examine the query-count exchange of this step's search now that the step is enqueued (host-side, no device synchronisation; the
        peers posted their counts at the start of their step) -- a mismatch surfaces here as a ValueError, before the caller synchronises on results whose
        collectives a peer never joined."""
        if self.database is not None and not torch.cuda.is_current_stream_capturing():
            self.database.check()

    def refine_stream(self, batches, query_scenes=None, patch_masks=None):
        """Software-pipelined ``refine`` over a sequence of batches (a generator of refined TSDFs, one per batch, in order).

        The front end of batch i + 1 -- query windows, query encoder, exact top-k, demotion, patch gather, and the U-Net backbone on the low-resolution
        input -- depends on nothing of batch i, is VALU / HBM work of ~2 ms and is issued on a helper stream BEFORE the back end of batch i
        (retrieval backbone, attention, decoder: LDS / MFMA work of ~7 ms on the caller's stream), so the two overlap on the GPU instead of following
        each other.  Same kernels on the same data as ``refine``: results are bit-identical; only the latency of a batch grows by one front end."""
        # the device / no_grad contexts are entered around the WORK of an iteration only and left before every yield: both are thread-global state, and a
        # generator that held them across its yields would run the consumer's loop body under no_grad on the engine's device (and keep them until it is
        # collected if the consumer breaks out)
        pending = None
        main = None
        state = {}

        def issue_front(raw, qs, pm, with_backbone=True):
            """front end of one batch on the helper stream: normalise -> (the chunk-level U-Net on its own stream) ; query encoder -> top-k -> demotion -> gather"""
            front = self._side_streams.get(('front', main.cuda_stream))
            if front is None:
                front = self._side_streams[('front', main.cuda_stream)] = torch.cuda.Stream(self.device)
            ready = torch.cuda.Event()
            ready.record(main)                                   # whatever produced `raw` on the caller's stream
            front.wait_event(ready)
            with torch.cuda.stream(front):
                raw.record_stream(front)
                x_in = self.normalise_input(raw)
                if with_backbone:
                    x_back, side = self._fork_backbone(x_in)     # ... the backbone beside the retrieval on its own stream, as in refine()
                patches, _ = self.retrieve(raw, qs, pm)
                if with_backbone:
                    front.wait_stream(side)
                done = torch.cuda.Event()
                done.record(front)
            return (patches, x_back, done) if with_backbone else (patches, x_in, done)

        front_at = self.front_at                                     # snapshot: the pending tuple's meaning depends on it
        if front_at not in ('start', 'decoders', 'unet_decoders'):
            raise ValueError("RefinementEngine.front_at must be 'start', 'decoders' or 'unet_decoders', got %r" % (front_at,))
        for i, raw in enumerate(batches):
            qs = query_scenes[i] if query_scenes is not None else None
            pm = patch_masks[i] if patch_masks is not None else None
            with torch.cuda.device(self.device), torch.no_grad():
                if main is None:
                    main = torch.cuda.current_stream(self.device)
                late_unet = front_at == 'unet_decoders'
                if pending is None or front_at in ('start', 'unet_decoders'):
                    nxt = issue_front(raw, qs, pm, with_backbone=not late_unet)
                    out = self._finish_pipelined(main, *pending, late_unet=late_unet) if pending is not None else None
                else:
                    # the next batch's front end is issued from INSIDE this batch's back end, behind its encoder launches: it then runs beside the decoder
                    # stages (MFMA-bound) instead of beside the first encoder layers (VALU / HBM-bound like the front end's own scan and gather)
                    def mid():
                        state['nxt'] = issue_front(raw, qs, pm)
                    out = self._finish_pipelined(main, *pending, after_encoders=mid)
                    nxt = state.pop('nxt')
                pending = nxt
                self._check_database()
            if out is not None:
                yield out
        if pending is not None:
            with torch.cuda.device(self.device), torch.no_grad():
                out = self._finish_pipelined(main, *pending, late_unet=front_at == 'unet_decoders')
            yield out

    def _finish_pipelined(self, main, patches, x_back, done, after_encoders=None, late_unet=False):
        main.wait_event(done)
        patches.record_stream(main)
        x_back.record_stream(main)
        if late_unet:
            # `x_back` is the normalised INPUT: the chunk-level U-Net of this batch is forked from inside its own back end, behind the encoder launches, and
            # joined in front of the attention -- its ~40 small launches then run beside the decoder stages (non-persistent launches of thousands of
            # workgroups, which absorb a borrowed CU) instead of beside the persistent first layers
            x_in, box = x_back, {}

            def fork():
                box['x_back'], box['side'] = self._fork_backbone(x_in)
            feats = self.retrieval_backbone(patches, after_encoders=fork)
            main.wait_stream(box['side'])
            return self._attend_and_decode(box['x_back'], feats, None)
        feats = self.retrieval_backbone(patches, after_encoders=after_encoders)
        return self._attend_and_decode(x_back, feats, None)

    @_on_engine_device
    def capture_graph(self, input_raw, query_scene=None, patch_mask=None):
        """Capture one whole refine() step for a FIXED batch shape into a HIP graph: -> (graph, static_input, static_output).
        Replaying costs one launch instead of ~150: the step of a single chunk is launch-bound (host enqueue ~2 ms against
        ~1.5 ms of GPU work).  Copy new chunks into ``static_input`` and ``graph.replay()``; the result is in ``static_output``.
        Weights must not be re-packed afterwards (load_state_dict -> capture again)."""
        static_in = input_raw.clone()
        cur = torch.cuda.current_stream(self.device)
        warm = torch.cuda.Stream(self.device)
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):                                # warm-up off the capture stream: packs weights, sizes workspaces
            for _ in range(2):
                self.refine(static_in, query_scene, patch_mask=patch_mask)
        cur.wait_stream(warm)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = self.refine(static_in, query_scene, patch_mask=patch_mask)
        return graph, static_in, static_out
