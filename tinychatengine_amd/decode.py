"""One decode token's worth of hot-path work: the linears of every transformer block + lm_head, on synthetic weights.

This is the caller side of the path as the reference issues it (llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu:73-115,
Int4llamaAttention.cu:116-229, Int4llamaForCausalLM.cu:17-50: per block qkv_proj, o_proj, gate_proj, up_proj, down_proj,
then lm_head) -- 5 x layers + 1 GEMV launches per token in the reference.  Here the linears that read the same
activation are issued as one grouped launch (q/k/v; gate/up) and the whole token is one hipGraph (capi.Plan).
Attention, norms, RoPE, SiLU are NOT part of the hot path and are not modelled: each linear reads a
synthetic activation buffer of the right shape.

Multi-GPU (no reference counterpart): every linear is sharded column-wise (output channels) over the ranks;
per transformer block the rank-local slices of the block output are joined by ONE RCCL all-gather over xGMI
(`gathers_per_block=1`, the north-star definition), or by the four gathers a dependency-faithful transformer needs
(`gathers_per_block=4`: after qkv/attention, o_proj, gate/up, down_proj).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch

from . import capi, quantize
from .linear import Linear_half_int4
from .matmul import _stream


@dataclass(frozen=True)
class ModelShape:
    name: str
    hidden: int
    qkv: tuple[int, ...]  # output widths of the projections that read the block input (fused or separate)
    ffn: int
    vocab: int
    layers: int


# llm/include/model.h:68-83.  "baseline-named" is the shape set BASELINE.json's configs[1] spells out for its
# "Llama-3-8B" row (4096x4096, 4096x11008 -- i.e. Llama-2-7B widths, fused 12288 qkv as the CUDA model launches it,
# Int4llamaAttention.cu:124-125); "llama3-8b" is the true Llama-3-8B set (GQA k/v 1024, FFN 14336, vocab 128256).
SHAPES = {
    "baseline-named": ModelShape("baseline-named (Llama-2-7B-shaped, as BASELINE.json spells it)", 4096, (12288,), 11008, 32000, 32),
    "llama3-8b": ModelShape("llama3-8b (true shapes, model.h:83)", 4096, (4096, 1024, 1024), 14336, 128256, 32),
    "llama2-13b": ModelShape("llama2-13b (model.h:72)", 5120, (15360,), 13824, 32000, 40),
    "tiny": ModelShape("tiny (tests)", 256, (256, 128, 128), 512, 1024, 2),
}


def _synthetic_linear(n: int, k: int, seed: int, device, group_size: int, rank: int, world: int, std: float = 0.02) -> Linear_half_int4:
    """W ~ N(0, std^2) fp32 [N][K] (std = 0.02: SURVEY section 8d), quantized with the reference's q4_6 recipe; rank keeps rows
    [rank*N/P, (rank+1)*N/P).  The matrix is defined in eight row blocks, each drawn from its own seed, so a rank generates
    ONLY the blocks it owns (1/P of the fp32 matrix, not all of it) and every world size in {1, 2, 4, 8} sees the same values."""
    n_loc = n // world
    blocks = 8 if (n % 8 == 0 and 8 % world == 0) else 1
    if blocks == 1:  # odd sizes (tests): draw everything, keep the slice
        g = torch.Generator(device=device).manual_seed(seed)
        w = torch.empty((n, k), dtype=torch.float32, device=device).normal_(0.0, std, generator=g)[rank * n_loc:(rank + 1) * n_loc]
    else:
        rows = n // blocks
        per_rank = blocks // world
        w = torch.empty((n_loc, k), dtype=torch.float32, device=device)
        for j in range(per_rank):
            g = torch.Generator(device=device).manual_seed(seed * 8 + rank * per_rank + j)
            w[j * rows:(j + 1) * rows].normal_(0.0, std, generator=g)
    lin = Linear_half_int4.from_float(w.contiguous(), group_size)
    del w
    return lin


class DecodeLinears:
    """All W4A16 linears of one decode token for rank `rank` of `world`, plus the launch list / plan to run them."""

    def __init__(self, shape: ModelShape, device="cuda", group_size: int = 128, rank: int = 0, world: int = 1,
                 m: int = 1, seed: int = 1234, layers: int | None = None, dataflow: bool = False, prepack: bool = False):
        """dataflow = False: every linear reads its own fixed synthetic activation vector (the launches are ordered by the
        stream only).  dataflow = True (world 1, M = 1): the linears FEED each other the way they do in the decoder --
        x -> qkv; o reads the q slice of qkv's output (the attention between them is not part of this path and same-sized);
        o -> gate, up; gate's output -> down (standing in for silu(gate) * up, same size); down -> the next block's qkv;
        the last down -> lm_head.  Same shapes, same bytes, same launch list; W ~ N(0, 1/K) instead of N(0, 0.02^2) so that the
        activations stay O(1) down the 129-linear chain.  This is the form TCE_PLAN_TAGGED needs: it orders the data flow."""
        self.shape, self.rank, self.world, self.m, self.group_size = shape, rank, world, m, group_size
        self.dataflow = dataflow
        if dataflow and (world != 1 or m != 1 or shape.qkv[0] < shape.hidden):
            raise ValueError("dataflow wiring: world 1, M = 1, and a first qkv linear at least `hidden` wide")
        self.device = torch.device(device)
        self._host_ok = self.device.type != "cuda"  # CPU tests of the sharding logic build descriptors for host tensors
        L = shape.layers if layers is None else layers
        self.n_layers = L
        h, f = shape.hidden, shape.ffn
        for n in (*shape.qkv, h, f, shape.vocab):
            if n % world or (n // world) % 16:
                raise ValueError(f"N={n} does not shard {world}-way into multiples of 16 rows")
        mk = lambda n, k, s: _synthetic_linear(n, k, seed + s, self.device, group_size, rank, world, std=(k ** -0.5 if dataflow else 0.02))
        self.blocks = []
        for li in range(L):
            s = li * 16
            self.blocks.append(dict(
                qkv=[mk(n, h, s + i) for i, n in enumerate(shape.qkv)],
                o=mk(h, h, s + 4), gate=mk(f, h, s + 5), up=mk(f, h, s + 6), down=mk(h, f, s + 7)))
        self.lm_head = mk(shape.vocab, h, 999_983)
        self.prepacked = bool(prepack)
        if prepack:  # load-time re-layout (tce_w4a16_prepack): the descriptors then carry the packed copy and the decode launches run on it (csrc/w4a16_gemv_i8.hip)
            for l in self.all_linears():
                l.prepack()
        # activations (fp16).  x ~ N(0,1); every buffer a linear READS is full width (replicated input).
        gx = torch.Generator(device=self.device).manual_seed(4321)
        rnd = lambda *sz: torch.empty(sz, dtype=torch.float32, device=self.device).normal_(0, 1, generator=gx).to(torch.float16)
        self.x = rnd(m, h)          # block input
        self.attn = rnd(m, h)       # attention output stand-in (input of o_proj)
        self.h2 = rnd(m, h)         # post-attention-norm stand-in (input of gate/up)
        self.act = rnd(m, f)        # silu(gate)*up stand-in (input of down_proj)
        W = world
        e = lambda n: torch.empty((m, n), dtype=torch.float16, device=self.device)
        self.out_qkv = [e(n // W) for n in shape.qkv]
        self.out_o, self.out_gate, self.out_up, self.out_down = e(h // W), e(f // W), e(f // W), e(h // W)
        self.logits = e(shape.vocab // W)
        if dataflow:
            self.attn, self.h2, self.act = self.out_qkv[0][:, :h], self.out_o, self.out_gate
        # gather targets (world > 1)
        if W > 1 or os.environ.get("TCE_FORCE_GATHER_BUFFERS"):
            self.g_qkv = [e(n) for n in shape.qkv]
            self.g_o, self.g_gate, self.g_up, self.g_down, self.g_logits = e(h), e(f), e(f), e(h), e(shape.vocab)

    # ---- descriptors ----
    def block_launches(self, li: int) -> list[list[capi.W4A16Desc]]:
        b = self.blocks[li]
        x = self.out_down if self.dataflow and li > 0 else self.x  # dataflow: the previous block's down_proj output
        return [
            [l.desc(x, o, allow_host=self._host_ok) for l, o in zip(b["qkv"], self.out_qkv)],
            [b["o"].desc(self.attn, self.out_o, allow_host=self._host_ok)],
            [b["gate"].desc(self.h2, self.out_gate, allow_host=self._host_ok), b["up"].desc(self.h2, self.out_up, allow_host=self._host_ok)],
            [b["down"].desc(self.act, self.out_down, allow_host=self._host_ok)],
        ]

    def token_launches(self, grouped: bool = True) -> list[list[capi.W4A16Desc]]:
        out = []
        for li in range(self.n_layers):
            out += self.block_launches(li)
        out.append([self.lm_head.desc(self.out_down if self.dataflow else self.x, self.logits, allow_host=self._host_ok)])
        if not grouped:  # one launch per linear, as the reference issues them
            out = [[d] for g in out for d in g]
        return out

    def all_linears(self) -> list[Linear_half_int4]:
        ls = []
        for b in self.blocks:
            ls += [*b["qkv"], b["o"], b["gate"], b["up"], b["down"]]
        return ls + [self.lm_head]

    def token_bytes(self) -> int:
        """Algorithmic HBM bytes of one token on THIS rank (SURVEY §8d formula, summed over its linears)."""
        return sum(capi.algorithmic_bytes(self.m, l.out_features, l.in_features, self.group_size) for l in self.all_linears())

    def token_launches_one_per_block(self) -> list[list[capi.W4A16Desc]]:
        """The one-gather-per-block form's launch list (replicated inputs: nothing inside a block depends on anything inside it): a rank's linears of a
        block as ONE group of linears that share nothing (tce_w4a16_forward_independent / TCE_PLAN_INDEPENDENT) -- layers + 1 launches per token instead of 4 x layers + 1."""
        if self.dataflow:
            raise ValueError("the data-flow wiring orders the linears of a block: they are not independent")
        return [[d for g in self.block_launches(li) for d in g] for li in range(self.n_layers)] + \
               [[self.lm_head.desc(self.x, self.logits, allow_host=self._host_ok)]]

    def make_plan(self, grouped: bool = True, tagged: bool = False, overlapped: bool = False, tuned: bool = False, one_launch_per_block: bool = False) -> capi.Plan:
        if one_launch_per_block:
            return capi.Plan(self.token_launches_one_per_block(), independent=True)
        return capi.Plan(self.token_launches(grouped), tagged=tagged, overlapped=overlapped, tuned=tuned)

    # ---- eager issue (used inside torch graph capture for world > 1, and by tests) ----
    @staticmethod
    def _hip_launch(group: list[capi.W4A16Desc]) -> None:
        st = _stream()
        capi.check(capi.w4a16_forward_group(group, st) if len(group) > 1 else capi.w4a16_forward(group[0], st))

    @staticmethod
    def _hip_launch_independent(group: list[capi.W4A16Desc]) -> None:
        capi.w4a16_forward_independent(group, _stream())

    def run_block(self, li: int, launch=None) -> None:
        launch = launch or self._hip_launch
        for g in self.block_launches(li):
            launch(g)

    def run_lm_head(self, launch=None) -> None:
        (launch or self._hip_launch)([self.lm_head.desc(self.x, self.logits, allow_host=self._host_ok)])

    def attach_peer_comm(self, exchange) -> None:
        """The peer-write gather of the C ABI (tce_comm / tce_allgather_f16, csrc/comm.hip) for this rank: creates the window,
        exchanges the 64-byte IPC handles through `exchange(bytes) -> list[bytes]` (e.g. torch.distributed.all_gather_object)
        and maps the peers.  run_token_distributed(gather="peer") then uses it instead of torch.distributed."""
        n_max = max(*self.shape.qkv, self.shape.hidden, self.shape.ffn, self.shape.vocab)
        # the window holds the widest gathered tensor of ALL M rows: tce_allgather_rows_f16 stages the ranks' whole [M][N/P] blocks through it (M = 64, hidden 4096 is
        # 262144 halves -- a one-row window would send every M > 1 exchange to RCCL, which this path does not set up).  Beyond 64 KiB slices the peer-write
        # kernel is slower than the links (csrc/comm.hip: RCCL takes those exchanges when the host called tce_comm_rccl_init), but it is correct at every size that fits.
        self.comm = capi.Comm(self.rank, self.world, n_max * self.m, slots=8)
        self.comm.connect(exchange(self.comm.export()))

    def _rows_workspace(self, full: torch.Tensor) -> torch.Tensor:
        """tce_allgather_rows_f16's workspace (M > 1): one buffer per rank, sized for the widest gathered tensor."""
        if getattr(self, "_rows_ws", None) is None or self._rows_ws.numel() < full.numel():
            n_max = max(*self.shape.qkv, self.shape.hidden, self.shape.ffn, self.shape.vocab)
            self._rows_ws = torch.empty(self.m * n_max, dtype=torch.float16, device=self.device)
        return self._rows_ws[:full.numel()]

    def check_comm(self) -> None:
        """Synchronises and raises if a peer-write gather of this rank ever gave up waiting (tce_comm_status): its outputs since then are
        void.  A host that uses the outputs calls this per token or per batch of tokens; after a host-side barrier tce_comm_reset re-arms."""
        if getattr(self, "comm", None) is not None and self.comm.status() != 0:
            raise RuntimeError(f"rank {self.rank}: a peer-write all-gather timed out (tce_comm_status != 0); the token's outputs are void")

    def run_token_distributed(self, gathers_per_block: int = 1, launch=None, gather: str = "rccl", check: bool = False, block_launch: str = "one") -> None:
        """world > 1: per block, the rank-local GEMVs on N/P shards, then the all-gather(s) of the fp16 output slices --
        gather="rccl": torch.distributed (backend 'nccl' == RCCL over xGMI on the GPU box; 'gloo' in the CPU tests, where
        `launch` is a CPU stand-in for the kernel launch); gather="peer": tce_allgather_f16, one peer-write kernel per
        exchange (attach_peer_comm first).  gathers_per_block = 1 is the north-star definition (all five linears of a block
        from replicated inputs, one gather of the block output); 4 is the dependency-faithful variant.
        block_launch (gathers_per_block = 1 only; round 6): "one" -- the rank's linears of a block are ONE launch (tce_w4a16_forward_independent: they share nothing and
        depend on nothing inside the block); "four" -- q/k/v grouped, o, gate/up grouped, down, as through round 5; "fused" (gather="peer" only) -- one launch per
        block WITH the exchange of the block output inside it (tce_w4a16_forward_independent_gather): layers + 1 launches per token and nothing else.  Same outputs, bit
        for bit."""
        if block_launch == "fused" and (gather != "peer" or gathers_per_block != 1 or self.m != 1):
            raise ValueError("block_launch='fused': the peer-write exchange, one gather per block, decode rows")
        launch_ind = launch or self._hip_launch_independent  # (a CPU stand-in handles any list of descriptors)
        launch = launch or self._hip_launch
        m = self.m
        if gather == "peer":
            st = _stream()
            slot_of = {}

            def ag(full, part):  # one slot per gather site: a site's epochs advance once per token on every rank
                slot = slot_of.setdefault(full.data_ptr(), len(slot_of))
                if m == 1:
                    self.comm.allgather(slot, part.data_ptr(), full.data_ptr(), full.numel(), st)
                else:  # M > 1 rows (a sharded prompt): the ranks' [M][N/P] blocks, rows laid side by side (peer-write kernel or RCCL by size, csrc/comm.hip)
                    ws = self._rows_workspace(full)
                    self.comm.allgather_rows(slot, part.data_ptr(), full.data_ptr(), m, full.shape[1], ws.data_ptr(), st)
        else:
            import torch.distributed as dist

            def ag(full, part):
                if m == 1:
                    dist.all_gather_into_tensor(full.view(-1), part.view(-1))
                else:  # all_gather of whole blocks is rank-major [P][M][N/P]; the rows go side by side afterwards
                    ws = self._rows_workspace(full).view(self.world, m, part.shape[1])
                    dist.all_gather_into_tensor(ws.view(-1), part.contiguous().view(-1))
                    full.copy_(ws.permute(1, 0, 2).reshape(m, full.shape[1]))
        for li in range(self.n_layers):
            lch = self.block_launches(li)
            if gathers_per_block == 1:
                if block_launch == "fused":  # the exchange inside the block's one launch (tce_w4a16_forward_independent_gather): layers + 1 launches per token
                    flat = [d for g in lch for d in g]
                    self.comm.forward_independent_gather(flat, len(flat) - 1, slot_of.setdefault(self.g_down.data_ptr(), len(slot_of)), self.g_down.data_ptr(), st)
                    continue
                if block_launch == "one":
                    launch_ind([d for g in lch for d in g])
                else:
                    for g in lch:
                        launch(g)
                ag(self.g_down, self.out_down)
            else:
                launch(lch[0])
                for gfull, part in zip(self.g_qkv, self.out_qkv):
                    ag(gfull, part)
                launch(lch[1]); ag(self.g_o, self.out_o)
                launch(lch[2]); ag(self.g_gate, self.out_gate); ag(self.g_up, self.out_up)
                launch(lch[3]); ag(self.g_down, self.out_down)
        if gathers_per_block == 1 and block_launch == "fused":
            self.comm.forward_independent_gather([self.lm_head.desc(self.x, self.logits, allow_host=self._host_ok)], 0, slot_of.setdefault(self.g_logits.data_ptr(), len(slot_of)),
                                                 self.g_logits.data_ptr(), st)
        else:
            launch([self.lm_head.desc(self.x, self.logits, allow_host=self._host_ok)])
            ag(self.g_logits, self.logits)
        if check and gather == "peer":  # (synchronises: not for a timed loop or a graph capture)
            self.check_comm()
