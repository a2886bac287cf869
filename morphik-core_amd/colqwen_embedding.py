"""MI355XColQwen2EmbeddingModel -- the reference's OWN encoder family behind the BaseEmbeddingModel surface.

core/embedding/colpali_embedding_model.py:47-52 loads `ColQwen2_5.from_pretrained("tsystems/colqwen2.5-3b-multilingual-v1.0")`
+ `ColQwen2_5_Processor` from colpali-engine (not vendored).  The same late-interaction head ships in transformers as
`ColQwen2ForRetrieval` / `ColQwen2Processor` (Qwen2-VL backbone, dynamic-resolution vision tower, 128-d projection, L2
normalisation): unlike ColPali's fixed 1030 rows, the number of rows PER PAGE depends on the page's resolution
(image_grid_thw), so pages arrive RAGGED -- exactly what the slab's fixed `stride_rows` slot + per-page `n_rows` is for.

Same surface as MI355XColpaliEmbeddingModel (embed_for_ingestion / embed_for_query / generate_embeddings /
embed_for_ingestion_device / latest_ingest_timing), same image-chunk recognition (colpali_embedding_model.py:83-100).
The forward runs on PyTorch-ROCm in bf16 (plumbing); rows stay on the GPU for mv_index_add_device.  A processor is
REQUIRED (the chat-template prompt and the patch grid are its job): pass a checkpoint directory, or a (model, processor)
pair -- random-init architectures with an offline-built processor are what the tests and this environment use.
"""
from __future__ import annotations

import asyncio
import logging
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .embedding import _decode_image
from .models import BaseEmbeddingModel, Chunk

logger = logging.getLogger(__name__)

PRESETS: Dict[str, Dict[str, Any]] = {
    # ColQwen2-v1.0 = Qwen2-VL-2B: ViT 32 x 1280 (patch 14, 2x2 merge) + Qwen2 1.5 B decoder
    "colqwen2-v1.0": dict(
        vision=dict(depth=32, embed_dim=1280, hidden_size=1536, num_heads=16, mlp_ratio=4, patch_size=14, spatial_merge_size=2, temporal_patch_size=2),
        text=dict(vocab_size=151936, hidden_size=1536, intermediate_size=8960, num_hidden_layers=28, num_attention_heads=12, num_key_value_heads=2,
                  max_position_embeddings=32768),
    ),
    "tiny": dict(
        vision=dict(depth=2, embed_dim=32, hidden_size=64, num_heads=4, mlp_ratio=2, patch_size=14, spatial_merge_size=2, temporal_patch_size=2),
        text=dict(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                  max_position_embeddings=2048),
    ),
    # the reference's model: tsystems/colqwen2.5-3b-multilingual-v1.0 = Qwen2.5-VL-3B backbone (windowed-attention ViT) + 128-d head
    "colqwen2.5-3b": dict(
        backbone="qwen2_5_vl",
        vision=dict(depth=32, hidden_size=1280, out_hidden_size=2048, intermediate_size=3420, num_heads=16, patch_size=14, spatial_merge_size=2,
                    temporal_patch_size=2, window_size=112, fullatt_block_indexes=[7, 15, 23, 31]),
        text=dict(vocab_size=151936, hidden_size=2048, intermediate_size=11008, num_hidden_layers=36, num_attention_heads=16, num_key_value_heads=2,
                  max_position_embeddings=128000),
    ),
    "tiny-2.5": dict(
        backbone="qwen2_5_vl",
        vision=dict(depth=2, hidden_size=32, out_hidden_size=64, intermediate_size=64, num_heads=4, patch_size=14, spatial_merge_size=2,
                    temporal_patch_size=2, window_size=56, fullatt_block_indexes=[1]),
        text=dict(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                  max_position_embeddings=2048),
    ),
}


def build_random_colqwen2(preset: str, tokenizer_ids: Dict[str, int], device, dtype, seed: int = 0):
    """Random-init ColQwen2ForRetrieval of a preset architecture; the special-token ids come from the processor's tokenizer."""
    import torch
    from transformers import ColQwen2Config, ColQwen2ForRetrieval

    p = PRESETS[preset]
    if p.get("backbone") == "qwen2_5_vl":  # ColQwen2.5: the same retrieval head over the Qwen2.5-VL backbone
        from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLConfig as Qwen2VLConfig
        from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLTextConfig as Qwen2VLTextConfig
        from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLVisionConfig as Qwen2VLVisionConfig
    else:
        from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLConfig, Qwen2VLTextConfig, Qwen2VLVisionConfig
    txt = dict(p["text"])
    txt["vocab_size"] = max(txt["vocab_size"], max(tokenizer_ids.values()) + 1)
    head_dim = txt["hidden_size"] // txt["num_attention_heads"]
    sec = head_dim // 2  # mrope sections (temporal, height, width) must sum to head_dim / 2
    txt["rope_parameters"] = {"rope_type": "default", "rope_theta": 1000000.0, "mrope_section": [sec - 2 * (sec // 3), sec // 3, sec // 3]}
    txt.update(bos_token_id=tokenizer_ids["eos"], eos_token_id=tokenizer_ids["eos"], pad_token_id=tokenizer_ids["eos"])
    vlm = Qwen2VLConfig(text_config=Qwen2VLTextConfig(**txt), vision_config=Qwen2VLVisionConfig(**p["vision"]),
                        image_token_id=tokenizer_ids["image"], video_token_id=tokenizer_ids["video"],
                        vision_start_token_id=tokenizer_ids["vision_start"], vision_end_token_id=tokenizer_ids["vision_end"])
    cfg = ColQwen2Config(vlm_config=vlm, embedding_dim=128)
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    try:
        torch.set_default_dtype(dtype)
        with torch.device(device):
            model = ColQwen2ForRetrieval(cfg)
    finally:
        torch.set_default_dtype(prev)
    return model.to(device).eval()


class MI355XColQwen2EmbeddingModel(BaseEmbeddingModel):
    def __init__(self, model_name_or_path: Optional[str] = None, model: Any = None, processor: Any = None, device: Optional[str] = None,
                 batch_size: int = 8, fused_ops: Optional[bool] = None):
        import torch

        self.torch = torch
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self.batch_size = int(batch_size)
        self._timing: Dict[str, Any] = {}
        self._decode_pool = None
        if model is not None and processor is not None:
            self.model, self.processor = model.to(self.device).eval(), processor
        elif model_name_or_path:
            from transformers import ColQwen2ForRetrieval, ColQwen2Processor

            self.model = ColQwen2ForRetrieval.from_pretrained(model_name_or_path, dtype=self.dtype).to(self.device).eval()
            self.processor = ColQwen2Processor.from_pretrained(model_name_or_path)
        else:
            raise ValueError("MI355XColQwen2EmbeddingModel needs a checkpoint directory or a (model, processor) pair")
        # RMSNorm and the gated-MLP activation as one HIP pass each (encoder_ops.py; MV_ENCODER_FUSED_OPS=0 keeps the framework's kernels)
        self.fused_ops = {"rmsnorm": 0, "gated_mlp": 0, "gelu_epilogue": 0}
        if self.device.type == "cuda" and fused_ops is not False:
            from . import encoder_ops

            if fused_ops or encoder_ops.enabled_by_env():
                self.fused_ops = encoder_ops.patch_encoder(self.model)

    # ------------------------------------------------------------------ forward
    def _forward(self, batch) -> Tuple[Any, Any]:
        torch = self.torch
        batch = {k: (v.to(self.device) if hasattr(v, "to") else v) for k, v in batch.items()}
        if "pixel_values" in batch:
            batch["pixel_values"] = batch["pixel_values"].to(self.dtype)
        with torch.inference_mode():
            out = self.model(**{k: batch[k] for k in ("input_ids", "attention_mask", "pixel_values", "image_grid_thw") if k in batch})
        return out.embeddings.to(self.dtype), batch["attention_mask"]  # [B, T, 128] L2-normalised rows; padded rows zero

    def _acc(self, key: str, dt: float) -> None:
        self._timing[key] = self._timing.get(key, 0.0) + dt

    def _embed(self, kind: str, items: Sequence[Any]) -> List[Any]:
        """-> one [n_tok_i, 128] bf16 device tensor per item; n_tok_i varies with the page's resolution."""
        torch = self.torch
        t0 = time.perf_counter()
        batch = self.processor(images=list(items), return_tensors="pt") if kind == "image" else self.processor(text=list(items), return_tensors="pt", padding=True)
        t1 = time.perf_counter()
        emb, mask = self._forward(batch)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t2 = time.perf_counter()
        self._acc(f"{kind}_process", t1 - t0)
        self._acc(f"{kind}_model", t2 - t1)
        return [emb[j][mask[j].bool()].contiguous() for j in range(emb.shape[0])]

    # ------------------------------------------------------------------ BaseEmbeddingModel
    def _ingest_device(self, chunks: Union[Chunk, List[Chunk]]):
        torch = self.torch
        self._timing = {}
        t_start = time.perf_counter()
        if isinstance(chunks, Chunk) or not isinstance(chunks, (list, tuple)):
            chunks = [chunks]
        if not chunks:
            return torch.zeros((0, 128), dtype=self.dtype, device=self.device), []
        decoded = [(_decode_image(c) if (c.metadata or {}).get("is_image") else None) for c in chunks]
        images = [(i, d) for i, d in enumerate(decoded) if d is not None]
        texts = [(i, c.content) for i, (c, d) in enumerate(zip(chunks, decoded)) if d is None]  # undecodable images fall back to text (:96-100)
        per_chunk: List[Any] = [None] * len(chunks)
        for kind, items in (("image", images), ("text", texts)):
            for b0 in range(0, len(items), self.batch_size):
                part = items[b0 : b0 + self.batch_size]
                for (i, _x), rows in zip(part, self._embed(kind, [x for _i, x in part])):
                    per_chunk[i] = rows
        n_rows = [int(t.shape[0]) for t in per_chunk]
        rows = torch.cat(per_chunk, 0).contiguous()
        self._timing.update(image_count=len(images), text_count=len(texts), chunk_count=len(chunks), total=time.perf_counter() - t_start)
        self._timing["process"] = self._timing.get("image_process", 0.0) + self._timing.get("text_process", 0.0)
        self._timing["model"] = self._timing.get("image_model", 0.0) + self._timing.get("text_model", 0.0)
        return rows, n_rows

    async def embed_for_ingestion(self, chunks: Union[Chunk, List[Chunk]]) -> List[np.ndarray]:
        rows, n_rows = await asyncio.to_thread(self._ingest_device, chunks)
        host = rows.to(self.torch.float32).cpu().numpy()  # the reference's output contract: float32 ndarrays (:290-292)
        out, o = [], 0
        for n in n_rows:
            out.append(host[o : o + n])
            o += n
        return out

    async def embed_for_ingestion_device(self, chunks: Union[Chunk, List[Chunk]]) -> Tuple[Any, List[int]]:
        """-> (bf16 tensor [sum n_tok, 128] ON THE DEVICE, rows per chunk -- ragged): feed the MI355X stores directly."""
        return await asyncio.to_thread(self._ingest_device, chunks)

    async def embed_for_query(self, text: str) -> np.ndarray:
        return await self.generate_embeddings(text)

    async def generate_embeddings(self, content: Any) -> np.ndarray:
        def run():
            rows = self._embed("text" if isinstance(content, str) else "image", [content])[0]
            return rows.to(self.torch.float32).cpu().numpy()

        return await asyncio.to_thread(run)

    def latest_ingest_timing(self) -> Dict[str, Any]:
        keys = ("sorting", "image_process", "image_model", "image_convert", "image_total", "text_process", "text_model", "text_convert",
                "text_total", "process", "model", "convert", "image_count", "text_count", "total", "chunk_count")
        t = dict(self._timing)
        t.setdefault("image_total", t.get("image_process", 0.0) + t.get("image_model", 0.0))
        t.setdefault("text_total", t.get("text_process", 0.0) + t.get("text_model", 0.0))
        return {k: t.get(k, 0.0) for k in keys}
