"""Model configuration for the PaDT hot path (mirrors the fields the reference reads from HF's config.json:
Qwen2.5-VL text/vision configs + ``vl_decoder`` + ``use_visual_prototype_projection``, padt.py:117-130)."""
from dataclasses import dataclass, field, asdict
from typing import Tuple


@dataclass
class VisionConfig:
    hidden_size: int = 1280
    depth: int = 32
    num_heads: int = 16
    intermediate_size: int = 3420
    patch_size: int = 14
    temporal_patch_size: int = 2
    in_channels: int = 3
    spatial_merge_size: int = 2
    window_size: int = 112
    fullatt_block_indexes: Tuple[int, ...] = (7, 15, 23, 31)
    out_hidden_size: int = 2048


@dataclass
class PaDTConfig:
    # LLM (Qwen2.5-VL text model)
    vocab_size: int = 151936
    hidden_size: int = 2048
    num_hidden_layers: int = 36
    num_attention_heads: int = 16
    num_key_value_heads: int = 2
    intermediate_size: int = 11008
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    tie_word_embeddings: bool = True
    vision_config: VisionConfig = field(default_factory=VisionConfig)
    # PaDT
    use_visual_prototype_projection: bool = True
    lora_r: int = 64
    vl_decoder: dict = field(default_factory=lambda: {"hidden_size": 1280, "intermediate_size": 3420, "num_heads": 16,
                                                      "use_mask_loss": True})
    # special ids (HF configuration_qwen2_5_vl.py:123-124,182-185)
    image_token_id: int = 151655
    vision_start_token_id: int = 151652
    eos_token_id: int = 151645
    pad_token_id: int = 151643

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def merge_unit(self) -> int:
        return self.vision_config.spatial_merge_size ** 2

    @property
    def patch_dim(self) -> int:
        v = self.vision_config
        return v.in_channels * v.temporal_patch_size * v.patch_size ** 2

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_hf_dict(cls, d: dict) -> "PaDTConfig":
        """Accepts the checkpoint's config.json (4.50 flat layout, or 5.x ``text_config`` nesting)."""
        t = d.get("text_config", d)
        v = d.get("vision_config", {})
        rope = t.get("rope_scaling") or t.get("rope_parameters") or d.get("rope_scaling") or {}
        vc = VisionConfig(
            hidden_size=v.get("hidden_size", 1280), depth=v.get("depth", 32), num_heads=v.get("num_heads", 16),
            intermediate_size=v.get("intermediate_size", 3420), patch_size=v.get("patch_size", 14),
            temporal_patch_size=v.get("temporal_patch_size", 2), in_channels=v.get("in_channels", v.get("in_chans", 3)),
            spatial_merge_size=v.get("spatial_merge_size", 2), window_size=v.get("window_size", 112),
            fullatt_block_indexes=tuple(v.get("fullatt_block_indexes", (7, 15, 23, 31))),
            out_hidden_size=v.get("out_hidden_size", t.get("hidden_size", 2048)))
        dec = dict(d.get("vl_decoder", {"hidden_size": 1280, "intermediate_size": 3420, "num_heads": 16}))
        dec.setdefault("use_mask_loss", True)
        return cls(
            vocab_size=t.get("vocab_size", 151936), hidden_size=t.get("hidden_size", 2048),
            num_hidden_layers=t.get("num_hidden_layers", 36), num_attention_heads=t.get("num_attention_heads", 16),
            num_key_value_heads=t.get("num_key_value_heads", 2), intermediate_size=t.get("intermediate_size", 11008),
            rms_norm_eps=t.get("rms_norm_eps", 1e-6), rope_theta=t.get("rope_theta", rope.get("rope_theta", 1e6)),
            mrope_section=tuple(rope.get("mrope_section", (16, 24, 24))),
            tie_word_embeddings=d.get("tie_word_embeddings", t.get("tie_word_embeddings", True)), vision_config=vc,
            use_visual_prototype_projection=d.get("use_visual_prototype_projection", True), vl_decoder=dec,
            image_token_id=d.get("image_token_id", 151655), vision_start_token_id=d.get("vision_start_token_id", 151652),
            eos_token_id=d.get("eos_token_id", 151645) if not isinstance(d.get("eos_token_id"), list) else d["eos_token_id"][0],
            pad_token_id=d.get("pad_token_id", 151643) or 151643)


def padt_pro_3b() -> PaDTConfig:
    return PaDTConfig()


def padt_pro_7b() -> PaDTConfig:
    return PaDTConfig(vocab_size=152064, hidden_size=3584, num_hidden_layers=28, num_attention_heads=28,
                      num_key_value_heads=4, intermediate_size=18944, tie_word_embeddings=False,
                      vision_config=VisionConfig(out_hidden_size=3584))


def small_test_config(layers: int = 2, vit_depth: int = 4) -> PaDTConfig:
    """Real head dims (ViT/decoder 80, LLM 128) with few heads/layers: fast GPU parity runs that still go through the
    same kernel template instantiations as PaDT_Pro_3B."""
    return PaDTConfig(
        vocab_size=2048, hidden_size=256, num_hidden_layers=layers, num_attention_heads=2, num_key_value_heads=1,
        intermediate_size=704, mrope_section=(16, 24, 24),
        vision_config=VisionConfig(hidden_size=160, depth=vit_depth, num_heads=2, intermediate_size=424,
                                   fullatt_block_indexes=(1, 3), out_hidden_size=256),
        lora_r=16, vl_decoder={"hidden_size": 160, "intermediate_size": 424, "num_heads": 2, "use_mask_loss": True},
        image_token_id=2001, vision_start_token_id=2002, eos_token_id=2003, pad_token_id=2004)
