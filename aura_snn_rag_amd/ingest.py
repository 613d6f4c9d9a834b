"""Memory-ingest helpers with the reference's signatures (``colab_l4_training.py:187-350``).

``store_custom_memory``, ``retrieve_custom_memories``, ``one_shot_memorize_text``,
``one_shot_memorize_and_generate``, ``ingest_jsonl_to_memory`` and ``ingest_csv_pairs_to_memory``
keep the reference's argument order, return values, id formats (``jsonl-<n>``, ``csv-<n>``,
``external-<ts>``, ``oneshot-<ts>``) and record-field handling, so they can replace the
originals in the training script.  ``ingest_feature_batches`` is the bulk path for the
"10 M one-shot writes" configuration: pre-computed feature rows go to the bank in chunks through
the batched write kernel instead of one model forward + one write per record.
"""
from __future__ import annotations

import csv
import json
import time
from typing import Iterable, Optional, Sequence, Tuple

import torch


def store_custom_memory(hippocampus, features: torch.Tensor, memory_id: Optional[str] = None):
    """Store an external feature vector; a 2-D input is mean-pooled over dim 0 first."""
    if hippocampus is None:
        return
    feat = features.detach()
    if feat.dim() == 2:
        feat = feat.mean(dim=0)
    if memory_id is None:
        memory_id = f"external-{int(time.time())}"
    hippocampus.create_episodic_memory(memory_id=memory_id, event_id=memory_id, features=feat)
    return memory_id


def retrieve_custom_memories(hippocampus, query_features: torch.Tensor,
                             location: Optional[torch.Tensor] = None, k: int = 5):
    if hippocampus is None:
        return []
    if query_features.dim() > 1:
        query_features = query_features.mean(dim=0)
    return hippocampus.retrieve_similar_memories(query_features, location=location, k=k)


def one_shot_memorize_text(text: str, tokenizer, model, hippocampus, device,
                           memory_id: Optional[str] = None):
    """Encode ``text`` and let the model's forward store its pooled embedding
    (``store_memory=True``), exactly as the reference drives it."""
    if hippocampus is None or model is None or tokenizer is None:
        return None
    max_len = getattr(getattr(model, "config", None), "max_seq_len", 256)
    ids = tokenizer.encode(text, return_tensors='pt', truncation=True, max_length=max_len).to(device)
    mem_id = memory_id or f"oneshot-{int(time.time())}"
    model.eval()
    with torch.no_grad():
        model(ids, prosody=None, use_memory=False, store_memory=True, memory_ids=[mem_id])
    return mem_id


def one_shot_memorize_and_generate(support_text: str, prompt: str, tokenizer, model, hippocampus,
                                   device, max_new_tokens: int = 40, temperature: float = 0.7) -> str:
    one_shot_memorize_text(support_text, tokenizer, model, hippocampus, device)
    model.eval()
    max_len = getattr(getattr(model, "config", None), "max_seq_len", 256)
    generated = tokenizer.encode(prompt, return_tensors='pt').to(device)
    with torch.no_grad():
        for _ in range(max_new_tokens):
            logits, _ = model(generated[:, -max_len:], use_memory=True, store_memory=False)
            probs = torch.softmax(logits[:, -1, :] / temperature, dim=-1)
            nxt = torch.multinomial(probs, num_samples=1)
            generated = torch.cat([generated, nxt], dim=1)
            if tokenizer.eos_token_id is not None and (nxt == tokenizer.eos_token_id).all():
                break
    return tokenizer.decode(generated[0].tolist(), skip_special_tokens=True)


def jsonl_record_text(obj) -> Optional[str]:
    """Text of one JSONL record: ``text``, or an (instruction/output), (prompt/completion) or
    (input/output) pair -- the field precedence of ``colab_l4_training.py:299-312``."""
    if isinstance(obj, str):
        return obj or None
    if not isinstance(obj, dict):
        return None
    if "text" in obj:
        return obj["text"] or None
    if "instruction" in obj and "output" in obj:
        return f"Instruction: {obj['instruction']}\nResponse: {obj.get('output', '')}"
    if "prompt" in obj and "completion" in obj:
        return f"Prompt: {obj['prompt']}\nCompletion: {obj.get('completion', '')}"
    if "input" in obj and "output" in obj:
        return f"Input: {obj['input']}\nOutput: {obj.get('output', '')}"
    return None


def ingest_jsonl_to_memory(path: str, tokenizer, model, hippocampus, device, max_items: int = 1000) -> int:
    if hippocampus is None or model is None or tokenizer is None:
        return 0
    stored = 0
    with open(path, "r", encoding="utf-8", errors="ignore") as f:
        for line in f:
            if stored >= max_items:
                break
            line = line.strip()
            if not line:
                continue
            try:
                obj = json.loads(line)
            except Exception:
                continue
            text = jsonl_record_text(obj)
            if not text:
                continue
            one_shot_memorize_text(text, tokenizer, model, hippocampus, device, memory_id=f"jsonl-{stored}")
            stored += 1
    return stored


def ingest_csv_pairs_to_memory(path: str, tokenizer, model, hippocampus, device,
                               max_items: int = 1000, delimiter: str = ",") -> int:
    if hippocampus is None or model is None or tokenizer is None:
        return 0
    stored = 0
    with open(path, "r", encoding="utf-8", errors="ignore") as f:
        for row in csv.reader(f, delimiter=delimiter):
            if stored >= max_items:
                break
            if len(row) < 2:
                continue
            q, a = row[0].strip(), row[1].strip()
            if not q and not a:
                continue
            one_shot_memorize_text(f"Question: {q}\nAnswer: {a}", tokenizer, model, hippocampus,
                                   device, memory_id=f"csv-{stored}")
            stored += 1
    return stored


def ingest_feature_batches(hippocampus, batches: Iterable[Tuple[Sequence[str], torch.Tensor]],
                           max_items: Optional[int] = None) -> int:
    """Bulk one-shot writes of pre-computed features: each batch is ``(ids, feats [n, D])`` in any
    float dtype (bf16 producers are widened to the bank's fp32 on the device)."""
    stored = 0
    for ids, feats in batches:
        if max_items is not None and stored + len(ids) > max_items:
            keep = max_items - stored
            ids, feats = ids[:keep], feats[:keep]
        if len(ids) == 0:
            break
        hippocampus.create_episodic_memories(list(ids), feats)
        stored += len(ids)
        if max_items is not None and stored >= max_items:
            break
    return stored
