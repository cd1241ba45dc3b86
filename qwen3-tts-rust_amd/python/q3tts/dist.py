"""Request sharding across GPUs.  Utterances are independent (own prompt, KV pages, sampler, decoder state;
/root/reference/src/tts/engine.rs:445-656 touches no cross-request state), so ranks never exchange data on the
per-frame path.  The ONE collective is the broadcast of a newly registered voice (speaker embedding 2048 f32 = 8 KB,
plus reference codes / ref-text ids for clone voices) from the rank that owns the voice file: RCCL over xGMI on GPUs
(backend "nccl"), gloo in the CPU tests.  It is latency-bound (~10-20 us), so it is a single flat broadcast."""
import numpy as np
import torch
import torch.distributed as dist


def shard_requests(n_requests, rank, world):
    """Round-robin request -> rank map: request i runs on rank i % world."""
    return [i for i in range(n_requests) if i % world == rank]


def broadcast_voice(spk_emb, audio_codes=None, ref_text_ids=None, src=0, device="cpu"):
    """Broadcasts a VoiceFile's numeric payload (voice_file.rs:5-22) from `src`.  Non-src ranks pass None."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return (np.asarray(spk_emb, np.float32), np.asarray(audio_codes if audio_codes is not None else [], np.int64),
                np.asarray(ref_text_ids if ref_text_ids is not None else [], np.int64))
    rank = dist.get_rank()
    hdr = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == src:
        hdr[0] = len(spk_emb)
        hdr[1] = 0 if audio_codes is None else len(audio_codes)
        hdr[2] = 0 if ref_text_ids is None else len(ref_text_ids)
    dist.broadcast(hdr, src=src)
    n_e, n_c, n_t = [int(x) for x in hdr.tolist()]
    # one payload message: f32 embedding bit-cast into the int64 stream would lose nothing but keep it simple: two tensors
    emb = torch.zeros(n_e, dtype=torch.float32, device=device)
    ints = torch.zeros(n_c + n_t, dtype=torch.int64, device=device)
    if rank == src:
        emb.copy_(torch.as_tensor(np.asarray(spk_emb, np.float32)))
        if n_c + n_t:
            ints.copy_(torch.as_tensor(np.concatenate([np.asarray(audio_codes if n_c else [], np.int64),
                                                       np.asarray(ref_text_ids if n_t else [], np.int64)])))
    dist.broadcast(emb, src=src)
    if n_c + n_t:
        dist.broadcast(ints, src=src)
    ints = ints.cpu().numpy()
    return emb.cpu().numpy(), ints[:n_c].copy(), ints[n_c:].copy()
