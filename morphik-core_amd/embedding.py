"""MI355XColpaliEmbeddingModel -- BaseEmbeddingModel plugin for the multi-vector encoder.

Drop-in for core/embedding/colpali_embedding_model.py:ColpaliEmbeddingModel (:25-337):
  embed_for_ingestion(chunks) -> List[np.ndarray (n_tok,128) float32]   (:66-218, contract asserted by
                                  core/tests/unit/test_colpali_embedding.py:56-59,72-77)
  embed_for_query(text)       -> np.ndarray (Q,128) float32             (:229-234)
  generate_embeddings(str | PIL.Image)                                  (:236-271; document_service.py:286)
  latest_ingest_timing()      -> dict with the reference's metric keys  (:198-227)
Image chunks are recognised exactly as the reference does (:83-100): metadata["is_image"], raw bytes in
metadata["_image_bytes"] preferred, else a data URI / base64 in chunk.content; a chunk that fails to decode is
embedded as text.

The encoder itself runs on PyTorch-ROCm (torch.cuda == HIP on the MI355X), bf16 under inference_mode -- it is
plumbing here, not a hand-written kernel: `transformers.ColPaliForRetrieval` (SigLIP-So400m/14 @448 -> 1024 patch
tokens + prompt tokens -> Gemma-2B -> 128-d projection -> L2 normalisation), the architecture BASELINE.json names.
No checkpoints and no network exist in the build environment, so without `model_name_or_path` the model is
RANDOM-INIT of that architecture (throughput and plumbing are real, retrieval quality is not) and text is
tokenised by a deterministic hash tokenizer; with a path, weights and processor are loaded from disk.

New on this side of the boundary (ingest-side fusion): `embed_for_ingestion_device` leaves
the bf16 rows on the GPU so MI355XMultiVectorStore can append them with mv_index_add_device -- no D2H -> fp32 ->
Python list -> H2D round trip (the reference does .to(float32).numpy() per page, :290-292).
"""
from __future__ import annotations

import asyncio
import base64
import io
import os
import logging
import time
import zlib
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .models import BaseEmbeddingModel, Chunk

logger = logging.getLogger(__name__)

PRESETS: Dict[str, Dict[str, Any]] = {
    # ColPali-v1.2 = PaliGemma-3B-mix-448: SigLIP-So400m/14 @ 448 px (32 x 32 = 1024 patches) + Gemma-2B
    "colpali-v1.2": dict(
        vision=dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16, image_size=448, patch_size=14),
        text=dict(hidden_size=2048, intermediate_size=16384, num_hidden_layers=18, num_attention_heads=8, num_key_value_heads=1, head_dim=256,
                  vocab_size=257216),
        projection_dim=2048, image_token_index=257152,
    ),
    # unit-test size (same code path, CPU friendly)
    "tiny": dict(
        vision=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14),
        text=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=1, head_dim=16,
                  vocab_size=1024),
        projection_dim=64, image_token_index=1023,
    ),
}

N_QUERY_AUGMENTATION_TOKENS = 10  # ColPali appends 10 buffer tokens to every query (colpali_engine process_queries)
IMAGE_PROMPT_TOKENS = 6           # "Describe the image." + BOS/newline: 1024 + 6 = 1030 rows per page


def _decode_image(chunk: Chunk):
    """-> PIL.Image or None.  Mirrors colpali_embedding_model.py:83-100."""
    try:
        from PIL import Image
    except Exception:  # noqa: BLE001
        return None
    md = chunk.metadata or {}
    raw = md.get("_image_bytes")
    try:
        if raw is None:
            content = chunk.content or ""
            if content.startswith("data:"):
                content = content.split(",", 1)[1]
            raw = base64.b64decode(content)
        img = Image.open(io.BytesIO(raw))
        img.load()
        return img.convert("RGB")
    except Exception as e:  # noqa: BLE001
        logger.error("Error processing image chunk, falling back to text: %s", e)
        return None


class HashTokenizer:
    """Deterministic stand-in used only when no processor files are available: word -> 2 + crc32(word) % (vocab-3)."""

    def __init__(self, vocab_size: int, reserved_top: int):
        self.n = max(vocab_size - reserved_top - 2, 8)

    def __call__(self, text: str) -> List[int]:
        words = text.strip().split() or [""]
        return [2 + (zlib.crc32(w.encode("utf-8")) % self.n) for w in words]


class MI355XColpaliEmbeddingModel(BaseEmbeddingModel):
    def __init__(
        self,
        model_name_or_path: Optional[str] = None,
        preset: str = "colpali-v1.2",
        device: Optional[str] = None,
        batch_size: int = 32,  # the reference uses 8 in cloud mode, 1 self-hosted (:61); measured here (model only):
                               # 8 -> 109 pages/s, 16 -> 126, 32 -> 140, 64 -> 145 (profiles/r1/embed_batch_probe.json)
        seed: int = 0,
        model: Any = None,
        fused_ops: Optional[bool] = None,  # None: on for a GPU unless MV_ENCODER_FUSED_OPS=0; False: the framework's own kernels
    ):
        import torch

        self.torch = torch
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"  # reference order: mps -> cuda -> cpu (:27)
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self.batch_size = int(batch_size)
        self._decode_pool = None
        self.processor = None
        self._timing: Dict[str, Any] = {}
        t0 = time.perf_counter()
        if model is not None:
            self.model = model
            self.random_init = False
        elif model_name_or_path:
            from transformers import ColPaliForRetrieval, ColPaliProcessor

            self.model = ColPaliForRetrieval.from_pretrained(model_name_or_path, dtype=self.dtype).to(self.device).eval()
            self.processor = ColPaliProcessor.from_pretrained(model_name_or_path)
            self.random_init = False
        else:
            self.model = self._build_random(preset, seed)
            self.random_init = True
        self._patch_embedding_as_gemm()
        # RMSNorm and the gated-MLP activation as one HIP pass each (encoder_ops.py; MV_ENCODER_FUSED_OPS=0 keeps the framework's kernels)
        self.fused_ops = {"rmsnorm": 0, "gated_mlp": 0, "gelu_epilogue": 0}
        self.tuned_gemms = False
        if self.device.type == "cuda" and fused_ops is not False:
            from . import encoder_ops

            if fused_ops or encoder_ops.enabled_by_env():
                self.fused_ops = encoder_ops.patch_encoder(self.model)
            # GEMM solutions tuned for this architecture's shapes on an MI355X (ignored by PyTorch on any other stack)
            self.tuned_gemms = encoder_ops.load_tuned_gemms()
        cfg = self.model.config.vlm_config
        self.image_size = int(cfg.vision_config.image_size)
        self.n_image_tokens = (self.image_size // int(cfg.vision_config.patch_size)) ** 2
        self.image_token_index = int(cfg.image_token_index)
        self.vocab_size = int(cfg.text_config.vocab_size)
        self.tokenizer = HashTokenizer(self.vocab_size, self.vocab_size - self.image_token_index)
        logger.info("MI355XColpaliEmbeddingModel ready on %s in %.1fs (random_init=%s, %d image tokens)", self.device,
                    time.perf_counter() - t0, self.random_init, self.n_image_tokens)

    # ------------------------------------------------------------------ construction
    def _build_random(self, preset: str, seed: int):
        import torch
        from transformers import ColPaliConfig, ColPaliForRetrieval, PaliGemmaConfig
        from transformers.models.gemma import GemmaConfig
        from transformers.models.siglip import SiglipVisionConfig

        p = PRESETS[preset]
        vis = SiglipVisionConfig(vision_use_head=False, **p["vision"])
        txt = GemmaConfig(**p["text"])
        pg = PaliGemmaConfig(vision_config=vis, text_config=txt, image_token_index=p["image_token_index"], projection_dim=p["projection_dim"],
                             hidden_size=p["text"]["hidden_size"], vocab_size=p["text"]["vocab_size"])
        cfg = ColPaliConfig(vlm_config=pg, embedding_dim=128)
        torch.manual_seed(seed)
        prev = torch.get_default_dtype()
        try:
            torch.set_default_dtype(self.dtype)  # allocate the 3 B parameters directly in bf16 on the device
            with torch.device(self.device):
                model = ColPaliForRetrieval(cfg)
        finally:
            torch.set_default_dtype(prev)
        return model.to(self.device).eval()

    def _patch_embedding_as_gemm(self) -> int:
        """SigLIP's patch embedding is a Conv2d with kernel == stride (non-overlapping patches): the same arithmetic as
        an unfold + ONE hipBLASLt GEMM [B*1024, 588] x [588, 1152].  Going through MIOpen instead costs a solver search at
        first use (8 trial runs of `naive_conv_ab_nonpacked_fwd_nhwc`, 80 ms each at batch 32 -- 40 % of the GPU time of a
        short run) for no steady-state gain.  The module
        and its parameters stay in place (checkpoints load unchanged), only its forward is replaced.
        -> number of modules rewired."""
        import types

        import torch.nn as nn
        import torch.nn.functional as F

        def gemm_forward(conv, x):
            B, C, H, W = x.shape
            p = conv.kernel_size[0]
            gh, gw = H // p, W // p
            cols = x.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p)
            y = F.linear(cols, conv.weight.reshape(conv.out_channels, -1), conv.bias)
            return y.transpose(1, 2).reshape(B, conv.out_channels, gh, gw)

        n = 0
        for m in self.model.modules():
            if (isinstance(m, nn.Conv2d) and m.kernel_size == m.stride and m.padding in ((0, 0), "valid") and m.dilation == (1, 1)
                    and m.groups == 1 and m.kernel_size[0] == m.kernel_size[1]):
                m.forward = types.MethodType(gemm_forward, m)
                n += 1
        return n

    # ------------------------------------------------------------------ preprocessing
    def _pixel_values(self, images: Sequence[Any]):
        """PIL images / HxWx3 uint8 arrays -> [B,3,S,S] in [-1,1] (SigLIP mean = std = 0.5)."""
        torch = self.torch
        arrs = []
        for im in images:
            if hasattr(im, "resize") and not isinstance(im, np.ndarray):
                im = np.asarray(im.convert("RGB").resize((self.image_size, self.image_size)))
            a = np.asarray(im)
            if a.shape[:2] != (self.image_size, self.image_size):
                from PIL import Image

                a = np.asarray(Image.fromarray(a.astype(np.uint8)).resize((self.image_size, self.image_size)))
            arrs.append(a)
        x = torch.from_numpy(np.stack(arrs)).to(self.device)
        return (x.permute(0, 3, 1, 2).to(torch.float32) / 127.5 - 1.0).to(self.dtype)

    def _forward(self, input_ids, attention_mask, pixel_values=None):
        torch = self.torch
        with torch.inference_mode():
            out = self.model(input_ids=input_ids, attention_mask=attention_mask, pixel_values=pixel_values)
        return out.embeddings  # [B, T, 128], rows L2-normalised, padded rows zero

    def _embed_images_device(self, images: Sequence[Any]):
        """-> (rows [B*T,128] bf16 on the device, T)"""
        torch = self.torch
        t0 = time.perf_counter()
        if self.processor is not None:
            batch = self.processor(images=list(images), return_tensors="pt").to(self.device)
            ids, mask, pv = batch["input_ids"], batch["attention_mask"], batch["pixel_values"].to(self.dtype)
        else:
            pv = self._pixel_values(images)
            B = pv.shape[0]
            prompt = torch.tensor(self.tokenizer("Describe the image .")[: IMAGE_PROMPT_TOKENS - 2] + [1, 1], device=self.device)
            prompt = prompt[:IMAGE_PROMPT_TOKENS]
            ids = torch.cat([torch.full((B, self.n_image_tokens), self.image_token_index, device=self.device), prompt.expand(B, -1)], 1)
            mask = torch.ones_like(ids)
        t1 = time.perf_counter()
        if self.tuned_gemms:
            from .encoder_ops import tuned_gemms

            with tuned_gemms(True):  # the page batches' GEMM shapes are the ones the selections were tuned for
                emb = self._forward(ids, mask, pv)
        else:
            emb = self._forward(ids, mask, pv)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t2 = time.perf_counter()
        self._acc("image_process", t1 - t0)
        self._acc("image_model", t2 - t1)
        return emb.to(self.dtype).contiguous(), mask

    def _embed_texts_device(self, texts: Sequence[str]):
        torch = self.torch
        t0 = time.perf_counter()
        if self.processor is not None:
            batch = self.processor(text=list(texts), return_tensors="pt", padding=True).to(self.device)
            ids, mask = batch["input_ids"], batch["attention_mask"]
        else:
            toks = [[1] + self.tokenizer(t) + [0] * N_QUERY_AUGMENTATION_TOKENS for t in texts]
            L = max(len(t) for t in toks)
            ids = torch.zeros((len(toks), L), dtype=torch.long, device=self.device)
            mask = torch.zeros((len(toks), L), dtype=torch.long, device=self.device)
            for i, t in enumerate(toks):
                ids[i, : len(t)] = torch.tensor(t, device=self.device)
                mask[i, : len(t)] = 1
        t1 = time.perf_counter()
        emb = self._forward(ids, mask)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t2 = time.perf_counter()
        self._acc("text_process", t1 - t0)
        self._acc("text_model", t2 - t1)
        return emb.to(self.dtype).contiguous(), mask

    def _acc(self, key: str, dt: float) -> None:
        self._timing[key] = self._timing.get(key, 0.0) + dt

    # ------------------------------------------------------------------ BaseEmbeddingModel
    async def embed_for_ingestion(self, chunks: Union[Chunk, List[Chunk]]) -> List[np.ndarray]:
        rows, n_rows = await asyncio.to_thread(self._ingest_device, chunks)
        t0 = time.perf_counter()
        host = rows.to(self.torch.float32).cpu().numpy()  # the reference's output contract: float32 ndarrays
        out, o = [], 0
        for n in n_rows:
            out.append(host[o : o + n])
            o += n
        self._acc("convert", time.perf_counter() - t0)
        return out

    async def embed_for_ingestion_device(self, chunks: Union[Chunk, List[Chunk]]) -> Tuple[Any, List[int]]:
        """-> (bf16 tensor [sum n_tok, 128] ON THE DEVICE, rows per chunk): feed MI355XMultiVectorStore directly."""
        return await asyncio.to_thread(self._ingest_device, chunks)

    def _ingest_device(self, chunks: Union[Chunk, List[Chunk]]):
        torch = self.torch
        self._timing = {}
        t_start = time.perf_counter()
        if isinstance(chunks, Chunk) or not isinstance(chunks, (list, tuple)):
            chunks = [chunks]
        if not chunks:
            return torch.zeros((0, 128), dtype=self.dtype, device=self.device), []
        images, image_pos, texts, text_pos = [], [], [], []

        def decode(c):  # PNG/JPEG decode + RGB + resize on a pool thread (PIL releases the GIL in all three)
            img = _decode_image(c) if (c.metadata or {}).get("is_image") else None
            if img is None or self.processor is not None:
                return img  # a real processor does its own resizing / normalisation
            return np.asarray(img.resize((self.image_size, self.image_size)))

        t_dec = time.perf_counter()
        if len(chunks) > 1:
            if self._decode_pool is None:
                from concurrent.futures import ThreadPoolExecutor

                self._decode_pool = ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1), thread_name_prefix="mv-decode")
            decoded = list(self._decode_pool.map(decode, chunks))
        else:
            decoded = [decode(chunks[0])]
        self._acc("image_process", time.perf_counter() - t_dec)
        for i, c in enumerate(chunks):
            img = decoded[i]
            if img is not None:
                images.append(img)
                image_pos.append(i)
            else:
                texts.append(c.content)
                text_pos.append(i)
        per_chunk: List[Any] = [None] * len(chunks)
        for b0 in range(0, len(images), self.batch_size):
            emb, mask = self._embed_images_device(images[b0 : b0 + self.batch_size])
            full = bool(mask.all())  # fixed-size pages: every row is valid, no per-chunk gather (and no per-chunk sync)
            for j in range(emb.shape[0]):
                per_chunk[image_pos[b0 + j]] = emb[j] if full else emb[j][mask[j].bool()]
        for b0 in range(0, len(texts), self.batch_size):
            emb, mask = self._embed_texts_device(texts[b0 : b0 + self.batch_size])  # text documents use the query template (:310)
            for j in range(emb.shape[0]):
                per_chunk[text_pos[b0 + j]] = emb[j][mask[j].bool()]
        n_rows = [int(t.shape[0]) for t in per_chunk]
        rows = torch.cat(per_chunk, 0).contiguous()
        self._timing.update(image_count=len(images), text_count=len(texts), chunk_count=len(chunks), total=time.perf_counter() - t_start)
        self._timing["process"] = self._timing.get("image_process", 0.0) + self._timing.get("text_process", 0.0)
        self._timing["model"] = self._timing.get("image_model", 0.0) + self._timing.get("text_model", 0.0)
        return rows, n_rows

    async def embed_for_query(self, text: str) -> np.ndarray:
        return await self.generate_embeddings(text)

    async def generate_embeddings(self, content: Any) -> np.ndarray:
        def run():
            if isinstance(content, str):
                emb, mask = self._embed_texts_device([content])
            else:
                emb, mask = self._embed_images_device([content])
            return emb[0][mask[0].bool()].to(self.torch.float32).cpu().numpy()

        return await asyncio.to_thread(run)

    def latest_ingest_timing(self) -> Dict[str, Any]:
        keys = ("sorting", "image_process", "image_model", "image_convert", "image_total", "text_process", "text_model", "text_convert",
                "text_total", "process", "model", "convert", "image_count", "text_count", "total", "chunk_count")
        t = dict(self._timing)
        t.setdefault("image_total", t.get("image_process", 0.0) + t.get("image_model", 0.0))
        t.setdefault("text_total", t.get("text_process", 0.0) + t.get("text_model", 0.0))
        return {k: t.get(k, 0.0) for k in keys}
