// pkv_cache.cpp — host block manager (no HIP in this file; see pkv_capi.cpp for the device side).
#include "pkv_cache.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <map>
#include <set>

namespace pkv {

// ---- fork prototype semantics ---------------------------------------------------------
int ToyBlockManager::allocateBlock() {
    for (size_t i = 0; i < used_.size(); ++i)
        if (!used_[i]) { used_[i] = true; return (int) i; }
    return evictBlock();
}
int ToyBlockManager::evictBlock() {
    for (size_t i = 0; i < used_.size(); ++i)
        if (used_[i]) { used_[i] = false; used_[i] = true; return (int) i; }  // cleared, then re-marked
    return -1;
}
void ToyBlockManager::freeBlock(int i) {
    if (i < 0 || (size_t) i >= used_.size()) return;
    used_[i] = false;
}

bool ToyKvManager::insert(int * block, int * slot) {
    auto first_free = [&](int * b, int * s) {
        for (size_t i = 0; i < blocks_.size(); ++i)
            for (int j = 0; j < cap_; ++j)
                if (!blocks_[i][j]) { *b = (int) i; *s = j; return true; }
        return false;
    };
    int b, s;
    if (!first_free(&b, &s)) {  // ensureFreeSlot
        blocks_.emplace_back(cap_, false);
        b = (int) blocks_.size() - 1;
        s = 0;
    }
    blocks_[b][s] = true;
    *block = b;
    *slot  = s;
    return true;
}
bool ToyKvManager::release(int block, int slot) {
    if (block < 0 || (size_t) block >= blocks_.size() || slot < 0 || slot >= cap_) return false;
    blocks_[block][slot] = false;
    return true;
}

ToyBlockKVCache::ToyBlockKVCache(size_t num_blocks, size_t block_size, size_t num_kv_heads, size_t head_size)
    : num_blocks_(num_blocks), be_(block_size * num_kv_heads * head_size), mgr_((int) num_blocks) {}
void ToyBlockKVCache::initCache() {
    key_.assign(num_blocks_ * be_, 0.0f);
    value_.assign(num_blocks_ * be_, 0.0f);
}
void ToyBlockKVCache::writeToCache(const float * k, const float * v, const std::vector<int> & map) {
    for (size_t i = 0; i < map.size(); ++i) {
        const int slot = map[i];
        if (slot < 0 || (size_t) slot >= num_blocks_ || ((size_t) slot + 1) * be_ > key_.size()) continue;
        std::copy(k + i * be_, k + (i + 1) * be_, key_.begin() + (size_t) slot * be_);
        std::copy(v + i * be_, v + (i + 1) * be_, value_.begin() + (size_t) slot * be_);
    }
}
void ToyBlockKVCache::swapBlocks(ToyBlockKVCache & o, const std::vector<int> & map) {
    for (size_t i = 0; i < map.size(); ++i) {
        const int j = map[i];
        if (i >= num_blocks_ || j < 0 || (size_t) j >= o.num_blocks_) continue;
        std::swap_ranges(key_.begin() + i * be_, key_.begin() + (i + 1) * be_, o.key_.begin() + (size_t) j * be_);
        std::swap_ranges(value_.begin() + i * be_, value_.begin() + (i + 1) * be_, o.value_.begin() + (size_t) j * be_);
    }
}
void ToyBlockKVCache::copyBlocks(const ToyBlockKVCache & src, const std::vector<int> & map) {
    for (int idx : map) {
        if (idx < 0 || (size_t) idx >= num_blocks_ || (size_t) idx >= src.num_blocks_) continue;
        std::copy(src.key_.begin() + (size_t) idx * be_, src.key_.begin() + ((size_t) idx + 1) * be_, key_.begin() + (size_t) idx * be_);
        std::copy(src.value_.begin() + (size_t) idx * be_, src.value_.begin() + ((size_t) idx + 1) * be_, value_.begin() + (size_t) idx * be_);
    }
}
void ToyBlockKVCache::splitKVCache(const std::vector<float> & kv, std::vector<float> & k, std::vector<float> & v) {
    const size_t h = kv.size() / 2;
    k.assign(kv.begin(), kv.begin() + h);
    v.assign(kv.begin() + h, kv.begin() + 2 * h);
}
bool ToyBlockKVCache::insertKVBlock(const float * k, const float * v, int * pb, int * slot) {
    // the prototype sizes storage BEFORE KVCacheManager::insert() may append a block and then writes out of bounds on the
    // insert that grows it (test_kv_cache.cpp:44); here storage follows the manager, the indices are the same
    if (!mgr_.insert(pb, slot)) return false;
    const size_t need = (size_t) mgr_.n_blocks() * num_blocks_ * be_;  // slot capacity == num_blocks
    if (key_.size() < need) { key_.resize(need, 0.0f); value_.resize(need, 0.0f); }
    const size_t at = ((size_t) *pb * num_blocks_ + (size_t) *slot) * be_;
    std::copy(k, k + be_, key_.begin() + at);
    std::copy(v, v + be_, value_.begin() + at);
    return true;
}
bool ToyBlockKVCache::freeKVBlock(int pb, int slot) { return mgr_.release(pb, slot); }

void toyBlockKVAttentionFwd(const float * q, size_t n, ToyBlockKVCache & c, const float * k, const float * v,
                            const std::vector<int> & map, float * out) {
    c.writeToCache(k, v, map);
    std::copy(q, q + n, out);
}
void toyBlockKVAttentionPagedFwd(const float * q, size_t n, ToyBlockKVCache & c, const float * k, const float * v, float * out,
                                 size_t page) {
    for (size_t s = 0; s < n; s += page) toyBlockKVAttentionFwd(q + s, std::min(page, n - s), c, k, v, {0}, out + s);
}
void toyBlockKVAttentionRealFwd(const float * q, size_t n, const float * k, const float * v, float * out) {
    for (size_t i = 0; i < n; ++i) {
        float sc[4], m = -3.0e38f, sum = 0.0f;
        for (int j = 0; j < 4; ++j) { sc[j] = q[i] * k[j]; m = std::max(m, sc[j]); }
        for (int j = 0; j < 4; ++j) { sc[j] = std::exp(sc[j] - m); sum += sc[j]; }
        out[i] = 0.0f;
        for (int j = 0; j < 4; ++j) out[i] += sc[j] / sum * v[j];
    }
}

// ---- paged cache ---------------------------------------------------------------------------------------------------
KvCachePaged::KvCachePaged(const pkv_cache_params & p)
    : p_(p), seqs_(p.n_seq_max > 0 ? p.n_seq_max : 1), ref_(p.n_blocks > 0 ? p.n_blocks : 0, 0), n_free_((int32_t) ref_.size()) {}

int32_t KvCachePaged::alloc_block(bool allow_evict) {
    if (n_free_ > 0) {
        for (size_t i = 0; i < ref_.size(); ++i)
            if (ref_[i] == 0) { ref_[i] = 1; --n_free_; return (int32_t) i; }
    }
    // Eviction trims EVERY owner of the victim — the caller's own sequence included: callers that hold cell or block
    // indices across this call must recompute them afterwards (find_slot, cow_block), or ask for allow_evict = false.
    if (p_.evict_when_full && allow_evict) {
        // fork BlockManager::evictBlock: the lowest allocated block is taken away from whoever holds it.
        // Its cells become holes in every owner and trim() drops it from their tables.
        for (size_t i = 0; i < ref_.size(); ++i) {
            if (ref_[i] == 0) continue;
            if (i < pinned_.size() && pinned_[i]) continue;   // holds tokens of the ubatch being placed (find_slot): never a victim of that same call
            const int32_t b = (int32_t) i;
            for (auto & s : seqs_) {
                bool hit = false;
                for (size_t lb = 0; lb < s.blocks.size(); ++lb) {
                    if (s.blocks[lb] != b) continue;
                    hit = true;
                    const size_t c0 = lb * p_.block_size, c1 = std::min(s.pos.size(), c0 + (size_t) p_.block_size);
                    for (size_t c = c0; c < c1; ++c) { s.pos[c] = -1; s.delta[c] = 0; }
                }
                if (hit) trim(s);
            }
            if (ref_[i] != 0) { n_free_ += 1; }  // defensive: nobody may still hold it
            ref_[i] = 1;
            --n_free_;
            return b;
        }
    }
    return -1;
}

void KvCachePaged::unref_block(int32_t b) {
    if (b < 0 || (size_t) b >= ref_.size() || ref_[b] == 0) return;
    if (--ref_[b] == 0) ++n_free_;
}

void KvCachePaged::drop_seq(Seq & s) {
    for (int32_t b : s.blocks) unref_block(b);
    s.blocks.clear();
    s.pos.clear();
    s.delta.clear();
}

void KvCachePaged::trim(Seq & s) {
    const size_t BS = (size_t) p_.block_size;
    // (1) wholly-empty logical blocks are removed from the table (later cells slide down by BS; their
    //     positions are explicit, so nothing else changes) — this is what replaces defrag moves
    for (size_t lb = s.blocks.size(); lb-- > 0;) {
        const size_t c0 = lb * BS, c1 = std::min(s.pos.size(), c0 + BS);
        bool empty = true;
        for (size_t c = c0; c < c1 && empty; ++c) empty = s.pos[c] < 0;
        if (!empty) continue;
        unref_block(s.blocks[lb]);
        s.blocks.erase(s.blocks.begin() + lb);
        s.pos.erase(s.pos.begin() + c0, s.pos.begin() + c1);
        s.delta.erase(s.delta.begin() + c0, s.delta.begin() + c1);
    }
    // (2) trailing holes
    while (!s.pos.empty() && s.pos.back() < 0) { s.pos.pop_back(); s.delta.pop_back(); }
}

bool KvCachePaged::cow_block(Seq & s, size_t lb, bool allow_evict) {
    const int32_t old = s.blocks[lb];
    if (ref_[old] <= 1) return true;
    const int32_t nb = alloc_block(allow_evict);
    if (nb < 0) return false;
    // alloc_block may have evicted a block of s itself (evict_when_full): the table can be shorter now, `old` can sit at
    // another index, or `old` can be the victim (then nb == old and s no longer holds it).  Look it up again.
    size_t at = s.blocks.size();
    if (nb != old)
        for (size_t i = 0; i < s.blocks.size(); ++i) if (s.blocks[i] == old) { at = i; break; }
    if (at == s.blocks.size() || ref_[old] <= 1) {   // nothing left to copy: hand the fresh block back
        unref_block(nb);
        return true;
    }
    pending.push_back({false, old, 0, nb, 0});
    unref_block(old);
    s.blocks[at] = nb;
    return true;
}

void KvCachePaged::clear() {
    for (auto & s : seqs_) drop_seq(s);
    std::fill(ref_.begin(), ref_.end(), 0);
    n_free_    = (int32_t) ref_.size();
    has_shift_ = false;
    pending.clear();
}

static inline void norm_range(int32_t & p0, int32_t & p1) {  // /ref/src/llama-kv-cache.cpp:176-182
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = INT32_MAX;
}

bool KvCachePaged::seq_rm(int32_t seq_id, int32_t p0, int32_t p1) {
    norm_range(p0, p1);
    auto rm = [&](Seq & s) {
        for (size_t c = 0; c < s.pos.size(); ++c)
            if (s.pos[c] >= p0 && s.pos[c] < p1) { s.pos[c] = -1; s.delta[c] = 0; }
        trim(s);
    };
    if (seq_id < 0) { for (auto & s : seqs_) rm(s); }
    else if (valid(seq_id)) rm(seqs_[seq_id]);
    return true;
}

void KvCachePaged::seq_cp(int32_t src, int32_t dst, int32_t p0, int32_t p1) {
    if (src == dst || !valid(src) || !valid(dst)) return;
    norm_range(p0, p1);
    Seq & a = seqs_[src];
    Seq & b = seqs_[dst];
    const size_t BS = (size_t) p_.block_size;
    // cells of src inside the range
    std::vector<size_t> idx;
    for (size_t c = 0; c < a.pos.size(); ++c)
        if (a.pos[c] >= p0 && a.pos[c] < p1) idx.push_back(c);
    if (idx.empty()) return;
    const bool prefix = idx.back() + 1 == idx.size();  // cells 0..n-1
    if (b.pos.empty() && prefix) {
        // share the blocks (the reference inserts dst into the src cells' seq_id sets, :285-292)
        const size_t n = idx.size(), nblk = (n + BS - 1) / BS;
        b.blocks.assign(a.blocks.begin(), a.blocks.begin() + nblk);
        for (int32_t blk : b.blocks) ++ref_[blk];
        b.pos.assign(a.pos.begin(), a.pos.begin() + n);
        b.delta.assign(a.delta.begin(), a.delta.begin() + n);
        // cells of src beyond n that live in the last shared block are not part of dst: nothing to do, dst's
        // logical length ends at n and the first append to that block by either owner copies it (find_slot).
        return;
    }
    // general case: append copies of the cells to dst (device cell copies queued for pkv_cache_update)
    const size_t need_cells = b.pos.size() + idx.size();
    size_t need_blocks = (need_cells + BS - 1) / BS - b.blocks.size();
    if (!b.pos.empty() && b.pos.size() % BS != 0 && ref_[b.blocks.back()] > 1) ++need_blocks;
    if ((int32_t) need_blocks > n_free_) return;  // no room: dst is left unchanged (a copy never evicts: src is being read)
    if (!b.pos.empty() && b.pos.size() % BS != 0 && !cow_block(b, b.blocks.size() - 1, false)) return;
    for (size_t c : idx) {
        const size_t i = b.pos.size();
        if (i / BS == b.blocks.size()) {
            const int32_t nb = alloc_block(false);
            if (nb < 0) return;
            b.blocks.push_back(nb);
        }
        pending.push_back({true, a.blocks[c / BS], (int32_t) (c % BS), b.blocks[i / BS], (int32_t) (i % BS)});
        b.pos.push_back(a.pos[c]);
        b.delta.push_back(a.delta[c]);
    }
}

void KvCachePaged::seq_keep(int32_t seq_id) {
    for (int32_t s = 0; s < (int32_t) seqs_.size(); ++s)
        if (s != seq_id) drop_seq(seqs_[s]);
}

void KvCachePaged::seq_add(int32_t seq_id, int32_t p0, int32_t p1, int32_t delta) {
    if (delta == 0 || !valid(seq_id)) return;
    norm_range(p0, p1);
    if (p0 == p1) return;
    Seq & s = seqs_[seq_id];
    const size_t BS = (size_t) p_.block_size;
    for (size_t c = 0; c < s.pos.size(); ++c) {
        if (s.pos[c] < p0 || s.pos[c] >= p1) continue;
        // the K rows of this cell will be rotated in place: make the block private first.  If the pool is full the
        // shift is applied to the shared block, which is the reference's behaviour for shared cells (:360-364); a shift
        // never evicts (the cell indices of this loop must stay valid).
        cow_block(s, c / BS, false);
        has_shift_ = true;
        s.pos[c] += delta;
        s.delta[c] += delta;
        if (s.pos[c] < 0) { s.pos[c] = -1; s.delta[c] = 0; }  // :366-373
    }
    trim(s);
}

void KvCachePaged::seq_div(int32_t seq_id, int32_t p0, int32_t p1, int d) {
    if (d == 1 || d == 0 || !valid(seq_id)) return;
    norm_range(p0, p1);
    if (p0 == p1) return;
    Seq & s = seqs_[seq_id];
    const size_t BS = (size_t) p_.block_size;
    for (size_t c = 0; c < s.pos.size(); ++c) {
        if (s.pos[c] < p0 || s.pos[c] >= p1) continue;
        cow_block(s, c / BS, false);
        has_shift_ = true;
        const int32_t old = s.pos[c];
        s.pos[c] /= d;                    // :414-418
        s.delta[c] += s.pos[c] - old;
    }
}

int32_t KvCachePaged::seq_pos_max(int32_t seq_id) const {
    int32_t r = 0;  // the reference starts from 0 as well (:432)
    if (!valid(seq_id)) return r;
    for (int32_t p : seqs_[seq_id].pos) r = std::max(r, p);
    return r;
}

int32_t KvCachePaged::get_n_tokens() const {  // sum over cells of |seq_id set| (:133-141)
    int32_t n = 0;
    for (const auto & s : seqs_)
        for (int32_t p : s.pos) n += p >= 0;
    return n;
}

int32_t KvCachePaged::get_used_cells() const {  // distinct occupied physical cells (`used`)
    std::set<int64_t> cells;
    for (const auto & s : seqs_)
        for (size_t c = 0; c < s.pos.size(); ++c)
            if (s.pos[c] >= 0) cells.insert((int64_t) s.blocks[c / p_.block_size] * p_.block_size + (int64_t) (c % p_.block_size));
    return (int32_t) cells.size();
}

bool KvCachePaged::seq_is_regular(int32_t sid) const {
    if (!valid(sid)) return true;
    const auto & pos = seqs_[sid].pos;
    for (size_t c = 0; c < pos.size(); ++c)
        if (pos[c] != (int32_t) c) return false;
    return true;
}

int KvCachePaged::find_slot(int n_tok, const int32_t * seq_id, const int32_t * pos, int32_t * out_block, int32_t * out_off) {
    const size_t BS = (size_t) p_.block_size;
    // pass 1: how many fresh blocks would this ubatch need?  (all-or-nothing like the reference, :654)
    std::vector<int32_t> add(seqs_.size(), 0);
    for (int t = 0; t < n_tok; ++t) {
        if (!valid(seq_id[t])) return PKV_ERR_INVALID;
        ++add[seq_id[t]];
    }
    int64_t need = 0;
    for (size_t s = 0; s < seqs_.size(); ++s) {
        if (!add[s]) continue;
        const Seq & q = seqs_[s];
        need += (int64_t) ((q.pos.size() + add[s] + BS - 1) / BS) - (int64_t) q.blocks.size();
        if (!q.pos.empty() && q.pos.size() % BS != 0 && ref_[q.blocks.back()] > 1) ++need;  // copy-on-write of the tail block
    }
    if (need > n_free_ && !p_.evict_when_full) return PKV_ERR_NO_SPACE;
    // evict_when_full: a block that has taken a token of THIS ubatch is pinned until the call returns — the lowest-allocated-first policy would
    // otherwise hand block 0 out again right after filling it: two tokens of one ubatch in the same cell (the store launch races on it) and the
    // earlier tokens' (block, offset) stale.  No victim left -> PKV_ERR_NO_SPACE with the cache as it was before the call.
    struct Unpin { std::vector<char> & p; ~Unpin() { p.clear(); } } unpin{pinned_};
    std::vector<Seq> seqs0; std::vector<int32_t> ref0; int32_t free0 = n_free_; const size_t pend0 = pending.size();
    if (p_.evict_when_full) { pinned_.assign(ref_.size(), 0); seqs0 = seqs_; ref0 = ref_; }
    auto fail = [&]() { if (p_.evict_when_full) { seqs_ = seqs0; ref_ = ref0; n_free_ = free0; pending.resize(pend0); } return PKV_ERR_NO_SPACE; };
    // pass 2: place the tokens in batch order
    for (int t = 0; t < n_tok; ++t) {
        Seq & q = seqs_[seq_id[t]];
        // With evict_when_full an allocation can take a block away from q itself (trim() then shortens q.pos / q.blocks),
        // so the slot is (re)computed from the sequence's CURRENT length after every allocation.
        for (int guard = 0;; ++guard) {
            if (guard > 4) return fail();
            const size_t lb0 = q.pos.size() / BS;
            if (lb0 == q.blocks.size()) {
                const int32_t nb = alloc_block();
                if (nb < 0) return fail();
                if (q.pos.size() / BS == q.blocks.size()) { q.blocks.push_back(nb); break; }
                unref_block(nb);            // q shrank into a partly filled block: that one takes the token
                continue;
            }
            if (ref_[q.blocks[lb0]] > 1) {
                if (!cow_block(q, lb0)) return fail();
                continue;                   // re-check: the copy may have evicted from q
            }
            break;
        }
        const size_t i = q.pos.size(), lb = i / BS;
        q.pos.push_back(pos[t]);
        q.delta.push_back(0);
        out_block[t] = q.blocks[lb];
        out_off[t]   = (int32_t) (i % BS);
        if (!pinned_.empty()) pinned_[q.blocks[lb]] = 1;
    }
    return PKV_OK;
}

void KvCachePaged::defrag() {
    // Holes that do not fill a whole block survive trim(); defrag rewrites such sequences densely into fresh
    // blocks (cell copies are queued for the device).  Sequences without holes are already "defragmented":
    // the block table makes physical contiguity irrelevant.
    const size_t BS = (size_t) p_.block_size;
    for (auto & s : seqs_) {
        size_t n_valid = 0;
        for (int32_t p : s.pos) n_valid += p >= 0;
        if (n_valid == s.pos.size()) continue;
        const size_t nblk = (n_valid + BS - 1) / BS;
        if ((int32_t) nblk > n_free_) continue;  // not enough scratch blocks: leave as is
        Seq d;
        for (size_t c = 0; c < s.pos.size(); ++c) {
            if (s.pos[c] < 0) continue;
            const size_t i = d.pos.size();
            if (i / BS == d.blocks.size()) d.blocks.push_back(alloc_block());
            pending.push_back({true, s.blocks[c / BS], (int32_t) (c % BS), d.blocks[i / BS], (int32_t) (i % BS)});
            d.pos.push_back(s.pos[c]);
            d.delta.push_back(s.delta[c]);
        }
        for (int32_t b : s.blocks) unref_block(b);
        s = std::move(d);
    }
}

void KvCachePaged::view(int32_t n_seq_max, int32_t * cell_pos, int32_t * cell_seqs, int32_t * stats) const {
    const int32_t BS = p_.block_size, n_cells = (int32_t) ref_.size() * BS;
    std::vector<int32_t> pos(n_cells, -1), cnt(n_cells, 0);
    if (cell_seqs) std::fill(cell_seqs, cell_seqs + (size_t) n_cells * n_seq_max, -1);
    int32_t token_count = 0;
    for (int32_t sid = 0; sid < (int32_t) seqs_.size(); ++sid) {   // ascending ids, like std::set iteration
        const Seq & s = seqs_[sid];
        for (size_t c = 0; c < s.pos.size(); ++c) {
            if (s.pos[c] < 0) continue;
            const int32_t i = s.blocks[c / BS] * BS + (int32_t) (c % BS);
            pos[i] = s.pos[c] + s.delta[c];
            if (cell_seqs && cnt[i] < n_seq_max) cell_seqs[(size_t) i * n_seq_max + cnt[i]] = sid;
            ++cnt[i];
            ++token_count;
        }
    }
    int32_t used = 0, cur = -1, best = 0, best_idx = -1;
    for (int32_t i = 0; i < n_cells; ++i) {
        if (cell_pos) cell_pos[i] = pos[i];
        if (cnt[i] > 0) {
            ++used;
            if (cur >= 0 && i - cur > best) { best = i - cur; best_idx = cur; }
            cur = -1;
        } else if (cur < 0) {
            cur = i;
        }
    }
    if (cur >= 0 && n_cells - cur > best) { best = n_cells - cur; best_idx = cur; }
    if (stats) { stats[0] = n_cells; stats[1] = token_count; stats[2] = used; stats[3] = best; stats[4] = best_idx; }
}

int KvCachePaged::state_cells(int32_t seq_id, std::vector<StateCell> & out) const {
    out.clear();
    if (has_shift_ || !pending.empty()) return PKV_ERR_INVALID;   // positions / rows not yet consistent on the device
    const int32_t BS = p_.block_size;
    if (seq_id >= 0) {
        if (!valid(seq_id)) return PKV_OK;                        // unknown sequence: no cells (has_seq_id is false everywhere)
        const Seq & s = seqs_[seq_id];
        for (size_t c = 0; c < s.pos.size(); ++c)
            if (s.pos[c] >= 0) out.push_back({s.blocks[c / BS], (int32_t) (c % BS), s.pos[c], {}});
        return PKV_OK;
    }
    std::map<int64_t, StateCell> cells;                           // flat index block * BS + off, ascending
    for (int32_t sid = 0; sid < (int32_t) seqs_.size(); ++sid) {
        const Seq & s = seqs_[sid];
        for (size_t c = 0; c < s.pos.size(); ++c) {
            if (s.pos[c] < 0) continue;
            const int32_t b = s.blocks[c / BS], o = (int32_t) (c % BS);
            StateCell & cell = cells[(int64_t) b * BS + o];
            cell.block = b; cell.off = o; cell.pos = s.pos[c];
            cell.seq_ids.push_back(sid);                          // ascending, like std::set<llama_seq_id> iteration
        }
    }
    for (auto & kv : cells) out.push_back(std::move(kv.second));
    return PKV_OK;
}

int KvCachePaged::state_restore(int32_t dest, const std::vector<int32_t> & pos, const std::vector<std::vector<int32_t>> & seq_ids,
                                std::vector<StateTarget> & targets) {
    targets.clear();
    const size_t n = pos.size(), BS = (size_t) p_.block_size;
    if (dest >= 0) {                                              // single sequence (:1004-1042)
        if (!valid(dest)) return PKV_ERR_INVALID;
        for (const auto & ids : seq_ids) if (!ids.empty()) return PKV_ERR_INVALID;   // "invalid seq_id-agnostic kv cell"
        seq_rm(dest, -1, -1);
        if (n == 0) return PKV_OK;
        std::vector<int32_t> sid(n, dest), blk(n), off(n);
        const int rc = find_slot((int) n, sid.data(), pos.data(), blk.data(), off.data());
        if (rc != PKV_OK) return rc;
        for (size_t i = 0; i < n; ++i) targets.push_back({(int32_t) i, blk[i], off[i]});
        return PKV_OK;
    }
    // whole cache (:1043-1090)
    for (const auto & ids : seq_ids)
        for (int32_t id : ids) if (!valid(id)) return PKV_ERR_INVALID;
    clear();
    std::vector<std::vector<int32_t>> list(seqs_.size());         // per sequence: its file cells, in file order
    for (size_t i = 0; i < n; ++i)
        for (int32_t id : seq_ids[i]) list[id].push_back((int32_t) i);
    std::map<std::vector<int32_t>, int32_t> made;                  // file cells of a block -> the physical block that holds them
    for (size_t sid = 0; sid < seqs_.size(); ++sid) {
        const auto & L = list[sid];
        Seq & s = seqs_[sid];
        for (size_t c0 = 0; c0 < L.size(); c0 += BS) {
            const std::vector<int32_t> key(L.begin() + c0, L.begin() + std::min(L.size(), c0 + BS));
            auto it = made.find(key);
            int32_t b;
            if (it != made.end()) { b = it->second; ++ref_[b]; }
            else {
                b = alloc_block(false);     // a restore never evicts what it has just placed
                if (b < 0) { clear(); return PKV_ERR_NO_SPACE; }
                made.emplace(key, b);
                for (size_t j = 0; j < key.size(); ++j) targets.push_back({key[j], b, (int32_t) j});
            }
            s.blocks.push_back(b);
        }
        for (int32_t i : L) { s.pos.push_back(pos[i]); s.delta.push_back(0); }
    }
    return PKV_OK;
}

void KvCachePaged::take_shifts(std::vector<int32_t> & blk, std::vector<int32_t> & off, std::vector<int32_t> & dlt) {
    std::set<int64_t> seen;
    for (auto & s : seqs_) {
        for (size_t c = 0; c < s.pos.size(); ++c) {
            if (s.delta[c] == 0) continue;
            const int32_t b = s.blocks[c / p_.block_size], o = (int32_t) (c % p_.block_size);
            if (seen.insert((int64_t) b * p_.block_size + o).second) {
                blk.push_back(b); off.push_back(o); dlt.push_back(s.delta[c]);
            }
            s.delta[c] = 0;
        }
    }
    has_shift_ = false;
}

}  // namespace pkv
