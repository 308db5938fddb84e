"""Stand-ins for the components the pipeline glue takes as arguments: tiny randomly initialised
Hugging Face text encoders, hash tokenizers and a one-layer VAE decoder."""
import torch


class HashTokenizer:
    """text -> fixed-length ids (a tokenizer's call signature, the kwargs the stage passes)."""

    def __init__(self, vocab: int):
        self.vocab = vocab

    def __call__(self, texts, max_length=16, padding="max_length", truncation=True, **_):
        ids, mask = [], []
        for t in texts:
            row = [1 + (sum(w.encode()) * 31 + i) % (self.vocab - 2) for i, w in enumerate(t.split())]
            row = row[:max_length]
            mask.append([1] * len(row) + [0] * (max_length - len(row)))
            ids.append(row + [0] * (max_length - len(row)))
        return {"input_ids": torch.tensor(ids), "attention_mask": torch.tensor(mask)}


def tiny_text_encoders(joint_dim: int, pooled_dim: int, seed: int = 0):
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel

    torch.manual_seed(seed)
    clip = CLIPTextModel(CLIPTextConfig(vocab_size=100, hidden_size=pooled_dim, intermediate_size=64,
                                        num_hidden_layers=1, num_attention_heads=2,
                                        max_position_embeddings=77, projection_dim=pooled_dim,
                                        bos_token_id=1, eos_token_id=2, pad_token_id=0)).eval()
    t5 = T5EncoderModel(T5Config(vocab_size=100, d_model=joint_dim, d_kv=16, d_ff=64, num_layers=1,
                                 num_heads=2)).eval()
    return [clip, t5], [HashTokenizer(100), HashTokenizer(100)]


class TinyVAE(torch.nn.Module):
    """decode: [B, 16, h, w] -> [B, 3, 8h, 8w]; scaling / shift factors of the Flux VAE config."""

    scaling_factor = 0.3611
    shift_factor = 0.1159

    def __init__(self, seed: int = 0):
        super().__init__()
        torch.manual_seed(seed)
        self.up = torch.nn.ConvTranspose2d(16, 3, kernel_size=8, stride=8)

    def decode(self, z):
        return (torch.tanh(self.up(z)),)
