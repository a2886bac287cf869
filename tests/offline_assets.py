"""Offline-built processor / checkpoint assets for the encoder tests: there is no network and no model hub here, so the
tokenizers are synthesised with the `tokenizers` library (word-level vocabularies) and the models are tiny random-init
architectures saved with save_pretrained -- enough to drive the checkpoint + processor code paths end to end."""
import numpy as np


def _word_tokenizer(specials, words, **special_kw):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    vocab = {s: i for i, s in enumerate(specials)}
    vocab.setdefault("<unk>", len(vocab))
    for w in words:
        vocab.setdefault(w, len(vocab))
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    return PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", **special_kw), vocab


WORDS = ("describe the image . question : query what is shown in this total revenue by quarter hello world second here user assistant "
         "synthetic number about tables and totals").split()


def colpali_processor(image_size=56, patch=14):
    from transformers import ColPaliProcessor, SiglipImageProcessor

    tok, vocab = _word_tokenizer(["<pad>", "<bos>", "<eos>", "<image>", "\n"], WORDS, bos_token="<bos>", eos_token="<eos>", pad_token="<pad>",
                                 additional_special_tokens=["<image>"])
    ip = SiglipImageProcessor(size={"height": image_size, "width": image_size}, image_seq_length=(image_size // patch) ** 2)
    return ColPaliProcessor(image_processor=ip, tokenizer=tok), vocab


def colpali_checkpoint(path, seed=0):
    """Tiny random-init ColPaliForRetrieval + offline processor saved to `path` (a from_pretrained-able directory)."""
    import torch
    from transformers import ColPaliConfig, ColPaliForRetrieval, PaliGemmaConfig
    from transformers.models.gemma import GemmaConfig
    from transformers.models.siglip import SiglipVisionConfig

    proc, vocab = colpali_processor()
    vis = SiglipVisionConfig(vision_use_head=False, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14)
    txt = GemmaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=1, head_dim=16, vocab_size=max(len(vocab), 64))
    pg = PaliGemmaConfig(vision_config=vis, text_config=txt, image_token_index=vocab["<image>"], projection_dim=64, hidden_size=64, vocab_size=txt.vocab_size)
    torch.manual_seed(seed)
    model = ColPaliForRetrieval(ColPaliConfig(vlm_config=pg, embedding_dim=128)).eval()
    model.save_pretrained(path)
    proc.save_pretrained(path)
    return path


QWEN_SPECIALS = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>", "<|vision_pad|>", "<|image_pad|>", "<|video_pad|>"]


def colqwen2_processor(min_tokens=4, max_tokens=16):
    from transformers import ColQwen2Processor, Qwen2VLImageProcessor

    tok, vocab = _word_tokenizer(QWEN_SPECIALS, WORDS, eos_token="<|endoftext|>", pad_token="<|endoftext|>", additional_special_tokens=QWEN_SPECIALS[1:])
    ip = Qwen2VLImageProcessor(patch_size=14, merge_size=2, min_pixels=min_tokens * 28 * 28, max_pixels=max_tokens * 28 * 28)
    ids = {"eos": vocab["<|endoftext|>"], "image": vocab["<|image_pad|>"], "video": vocab["<|video_pad|>"],
           "vision_start": vocab["<|vision_start|>"], "vision_end": vocab["<|vision_end|>"]}
    return ColQwen2Processor(image_processor=ip, tokenizer=tok), ids


def page_image(rng, h, w):
    from PIL import Image

    img = rng.integers(200, 255, (h, w, 3), dtype=np.uint8)
    for _ in range(6):
        y, x = int(rng.integers(0, max(h - 8, 1))), int(rng.integers(0, max(w - 30, 1)))
        img[y : y + 4, x : x + 24] = rng.integers(0, 60)
    return Image.fromarray(img)


def png_bytes(img) -> bytes:
    import io

    buf = io.BytesIO()
    img.save(buf, format="PNG")
    return buf.getvalue()
