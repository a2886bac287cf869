"""Oracle-backed stand-in for MvIndex, for CPU tests of the HOST logic only (store bookkeeping,
sharded merge).  Test infrastructure: lives under tests/, never imported by the product."""
import numpy as np

from oracle import oracle as orc


class OracleIndex:
    def __init__(self, capacity_pages, stride_rows, device=0, id_base=0, mode="float", fde=None, with_binary=False, with_float_lo=False, **_):
        self.capacity, self.stride_rows, self.id_base, self.mode = capacity_pages, stride_rows, id_base, mode
        self.device = device
        self.float_lo = bool(with_float_lo)  # MV_WITH_FLOAT_LO: fp32 pages keep their low bits (hi + lo); else the slab holds bf16(x)
        if with_binary and mode == "float":
            self.mode = "binary"  # built by ShardedIndex from slab flags
        self.pages, self.ords, self.alive = [], [], []
        self.fde = fde  # oracle FdeConfig; enables mode "fde" / "fde_then_float"
        self.ext_fde = {}  # page -> caller-supplied document FDE (import_fde): replaces the oracle's own encoding of that page

    def __len__(self):
        return len(self.pages)

    def add(self, pages, doc_ordinals=None):
        first = len(self.pages)
        if first + len(pages) > self.capacity:
            raise RuntimeError("slab full")
        for i, p in enumerate(pages):
            p = np.asarray(p)
            f = orc.bf16_to_f32(p) if p.dtype == np.uint16 else np.asarray(p, np.float32).reshape(-1, 128)
            self.pages.append(f)
            self.ords.append(0 if doc_ordinals is None else int(doc_ordinals[i]))
            self.alive.append(True)
        return first

    def import_fde(self, page0, fde):
        fde = np.asarray(fde, np.float32)
        for i, v in enumerate(fde.reshape(len(fde), -1)):
            self.ext_fde[int(page0) + i] = v

    def remove_page(self, page):
        self.alive[page] = False

    def remove_doc(self, o):
        n = 0
        for i, x in enumerate(self.ords):
            if x == o and self.alive[i]:
                self.alive[i] = False
                n += 1
        return n

    def _mask(self, allow):
        m = np.array(self.alive, bool)
        if allow is not None:
            o = np.array(self.ords, dtype=np.int64)
            ok = (o < allow.size * 32) & (((allow[np.minimum(o >> 5, allow.size - 1)] >> (o & 31).astype(np.uint32)) & 1) == 1)
            m &= ok
        return m

    def score_all(self, q, mode=None, allow=None, q_fde=None):
        mode = mode or self.mode
        q = np.asarray(q)
        qf = orc.bf16_to_f32(q) if q.dtype == np.uint16 else np.asarray(q, np.float32)
        out = np.full(len(self.pages), -np.inf, np.float32)
        m = self._mask(allow)
        if mode == "fde":
            fq = np.asarray(q_fde, np.float32).reshape(-1) if q_fde is not None else orc.fde_encode(self.fde, orc.bf16_to_f32(orc.f32_to_bf16(qf)), True)
            slab = np.stack([self.ext_fde[i] if i in self.ext_fde else orc.fde_encode(self.fde, orc.bf16_to_f32(orc.f32_to_bf16(p)), False)
                             for i, p in enumerate(self.pages)])
            s = orc.fde_coarse_scores(fq, orc.f32_to_bf16(slab), use_cosine=True)
            out[m] = s[m]
            return out
        for i, p in enumerate(self.pages):
            if not m[i]:
                continue
            if mode == "binary":
                out[i] = orc.maxsim_binary(orc.sign_pack(p) if len(p) else np.zeros((0, 16), np.uint8), orc.sign_pack(qf))
            else:
                # the library's rule: an fp32 query is scored exactly (hi + lo halves); pages are bf16 unless the index keeps the lo slab
                out[i] = orc.maxsim_f32(qf, self._slab_page(p))
        return out

    def query(self, q, k, mode=None, allow=None, want_stats=False, coarse_n=None, q_fde=None):
        if (mode or self.mode) == "fde_then_float":
            # reference pipeline (fast_multivector_store.py:521-556): coarse top-n -> pad-to-longest rerank -> top-k
            n = coarse_n or min(10 * k, 75)
            cs, ci = orc.topk(self.score_all(q, "fde", allow, q_fde=q_fde), n)
            ci = ci[np.isfinite(cs)]
            if ci.size == 0:
                return np.zeros(0, np.float32), np.zeros(0, np.int64)
            sc = self.score_candidates(q, ci, pad_to=-1)  # every batch of 128 candidates pads to its own longest page
            order = np.lexsort((np.arange(ci.size), -sc.astype(np.float64)))[:k]  # ties keep coarse rank order
            return sc[order], ci[order] + self.id_base
        s = self.score_all(q, mode, allow, q_fde=q_fde)
        sc, ids = orc.topk(s, k)
        return sc, ids + self.id_base

    def page_rows(self, pages):
        return np.array([len(self.pages[int(i)]) for i in pages], np.int32)

    def score_candidates(self, q, cand, pad_to=0, pads=None):
        """pad_to = -1: the reference rule (score_multi_vector scores passages in batches of 128, each zero-padded to its
        own longest page); pads = explicit pad length per candidate."""
        q = np.asarray(q)
        qf = orc.bf16_to_f32(q) if q.dtype == np.uint16 else np.asarray(q, np.float32)
        cand = [int(i) for i in cand]
        if pads is None:
            if pad_to < 0:
                rows = [len(self.pages[i]) for i in cand]
                pads = [max(rows[j - j % 128 : j - j % 128 + 128]) for j in range(len(cand))]
            else:
                pads = [pad_to] * len(cand)
        return np.array([orc.maxsim_f32(qf, self._slab_page(self.pages[i]), int(pd)) for i, pd in zip(cand, pads)], np.float32)

    def _slab_page(self, p):
        return p if self.float_lo else orc.bf16_to_f32(orc.f32_to_bf16(p))

    def compact(self):
        o2n, pages, ords = [], [], []
        for p, o, a in zip(self.pages, self.ords, self.alive):
            o2n.append(len(pages) if a else -1)
            if a:
                pages.append(p)
                ords.append(o)
        self.ext_fde = {o2n[i]: v for i, v in self.ext_fde.items() if o2n[i] >= 0}
        self.pages, self.ords, self.alive = pages, ords, [True] * len(pages)
        return np.array(o2n, np.int64)

    def query_batch(self, queries, k, mode=None, allow=None, want_stats=False, allows=None, n_docs=0, q_fdes=None):
        out = []
        for j, q in enumerate(queries):
            a = allow if allows is None else allows[j]
            out.append(self.query(q, k, mode, None if a is None else np.asarray(a, np.uint32), q_fde=None if q_fdes is None else np.asarray(q_fdes)[j]))
        return out

    def save(self, path):
        import pickle

        with open(path + ".tmp", "wb") as f:
            pickle.dump({"capacity": self.capacity, "stride_rows": self.stride_rows, "id_base": self.id_base, "mode": self.mode,
                         "pages": self.pages, "ords": self.ords, "alive": self.alive, "ext_fde": self.ext_fde, "fde": self.fde}, f)
        import os

        os.replace(path + ".tmp", path)

    @classmethod
    def load(cls, path, device=0):
        import pickle

        with open(path, "rb") as f:
            d = pickle.load(f)
        self = cls(d["capacity"], d["stride_rows"], device=device, id_base=d["id_base"], mode=d["mode"])
        self.pages, self.ords, self.alive = d["pages"], d["ords"], d["alive"]
        self.ext_fde, self.fde = d.get("ext_fde", {}), d.get("fde")
        return self

    def close(self):
        pass


class OracleComm:
    """Stand-in for index.ShardComm over OracleIndex shards: per-shard top-k merged on the host with the library's rule
    (score desc; ties: shard asc, position asc == ascending global id).  Single-stage modes only."""

    def __init__(self, shards, transport="auto"):
        self.shards, self.transport = list(shards), "host"

    def close(self):
        pass

    def query(self, q, k, mode="float", allow=None, want_stats=False, q_fde=None):
        s_all, i_all = [], []
        for sh in self.shards:
            s, i = sh.query(q, k, mode=mode, allow=allow, q_fde=q_fde)
            ok = np.isfinite(s)
            s_all.append(s[ok])
            i_all.append(i[ok])
        s = np.concatenate(s_all) if s_all else np.zeros(0, np.float32)
        i = np.concatenate(i_all) if i_all else np.zeros(0, np.int64)
        order = np.lexsort((i, -s.astype(np.float64)))[:k]
        res = (s[order].astype(np.float32), i[order].astype(np.int64))
        return res + ([None] * len(self.shards),) if want_stats else res
