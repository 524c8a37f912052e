// pkv_cache.h — host block manager of the paged KV cache (C++ mirror of the llama_kv_cache method set).
//
// Replaces llama_kv_cache_unified's cell ring (/ref/src/llama-kv-cache.h:63-178, /ref/src/llama-kv-cache.cpp)
// with a vLLM-style logical->physical block table.  Allocation policy and the toy classes follow the fork
// prototype recovered in SURVEY.md §8(a-toy) (/ref/tests/test_block_manager.cpp, test_kv_cache.cpp).
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

#include "../../include/mi355_paged_kv.h"

namespace pkv {

// ---- fork prototype: BlockManager (test_block_manager.cpp) ------------------------------------------------
class ToyBlockManager {
  public:
    explicit ToyBlockManager(int n) : used_(n > 0 ? n : 0, false) {}
    int  allocateBlock();            // lowest free index, else evictBlock(), else -1
    int  evictBlock();               // lowest ALLOCATED index: cleared, then handed out again
    void freeBlock(int i);           // bounds-checked, idempotent
    const std::vector<bool> & getBlockStatus() const { return used_; }
  private:
    std::vector<bool> used_;
};

// ---- fork prototype: PhysicalBlock + KVCacheManager (test_kv_cache.cpp) --------------------------------------
class ToyKvManager {
  public:
    explicit ToyKvManager(int slot_cap) : cap_(slot_cap > 0 ? slot_cap : 1), blocks_(1, std::vector<bool>(cap_, false)) {}
    bool insert(int * block, int * slot);  // ensureFreeSlot(): append a block iff none has a free slot; first fit
    bool release(int block, int slot);     // bounds-checked
    int  n_blocks() const { return (int) blocks_.size(); }
  private:
    int                            cap_;
    std::vector<std::vector<bool>> blocks_;
};

// ---- fork prototype: BlockKVCache (test_kv_cache.cpp, test_block_kv_cache.cpp) + the block_kv_attention toys -----
// float-only host storage, exactly the prototype's semantics; be = block_size*heads*head_size
class ToyBlockKVCache {
  public:
    ToyBlockKVCache(size_t num_blocks, size_t block_size, size_t num_kv_heads, size_t head_size);
    void initCache();                                                          // zero-fill num_blocks * be
    void writeToCache(const float * k, const float * v, const std::vector<int> & slot_mapping);  // block i of input -> slot map[i]
    void swapBlocks(ToyBlockKVCache & other, const std::vector<int> & mapping);    // this block i <-> other block map[i]
    void copyBlocks(const ToyBlockKVCache & src, const std::vector<int> & mapping); // same-index copy src[idx] -> this[idx]
    static void splitKVCache(const std::vector<float> & kv, std::vector<float> & k, std::vector<float> & v);
    bool insertKVBlock(const float * k, const float * v, int * physical_block, int * slot);
    bool freeKVBlock(int physical_block, int slot);
    const std::vector<float> & getKeyCache() const { return key_; }
    const std::vector<float> & getValueCache() const { return value_; }
    size_t blockElems() const { return be_; }
  private:
    size_t num_blocks_, be_;
    std::vector<float> key_, value_;
    ToyKvManager mgr_;
};
// blockKVAttentionFwd: writeToCache + copy(query -> output); PagedFwd: page by page with mapping {0};
// RealFwd: per query scalar q_i, scores q_i*k[j] (j < 4), max-subtracted softmax, out_i = sum p_j v[j]
void toyBlockKVAttentionFwd(const float * q, size_t n, ToyBlockKVCache & c, const float * k, const float * v,
                            const std::vector<int> & slot_mapping, float * out);
void toyBlockKVAttentionPagedFwd(const float * q, size_t n, ToyBlockKVCache & c, const float * k, const float * v, float * out,
                                 size_t page_size);
void toyBlockKVAttentionRealFwd(const float * q, size_t n, const float * k, const float * v, float * out);

// ---- the paged cache ------------------------------------------------------------------------------------------
// device work queued by host edits, executed IN ORDER by pkv_cache_update: whole-block copies (copy-on-write)
// and single-cell copies (seq_cp of unaligned ranges, defrag)
struct PendingOp { bool cell; int32_t src_block, src_off, dst_block, dst_off; };

class KvCachePaged {
  public:
    explicit KvCachePaged(const pkv_cache_params & p);

    // llama_memory_i / llama_kv_cache (/ref/src/llama-memory.h:7-21, /ref/src/llama-kv-cache.h:17-26)
    void    clear();
    void    defrag();
    bool    seq_rm(int32_t seq_id, int32_t p0, int32_t p1);
    void    seq_cp(int32_t src, int32_t dst, int32_t p0, int32_t p1);
    void    seq_keep(int32_t seq_id);
    void    seq_add(int32_t seq_id, int32_t p0, int32_t p1, int32_t delta);
    void    seq_div(int32_t seq_id, int32_t p0, int32_t p1, int d);
    int32_t seq_pos_max(int32_t seq_id) const;
    int32_t get_n_tokens() const;
    int32_t get_used_cells() const;
    bool    get_can_shift() const { return true; }

    // find_slot (/ref/src/llama-kv-cache.cpp:453-672): all-or-nothing placement of a ubatch
    int find_slot(int n_tok, const int32_t * seq_id, const int32_t * pos, int32_t * out_block, int32_t * out_off);

    // introspection
    int32_t seq_len(int32_t s) const { return valid(s) ? (int32_t) seqs_[s].pos.size() : 0; }
    const std::vector<int32_t> & seq_blocks(int32_t s) const { return seqs_[s].blocks; }
    const std::vector<int32_t> & seq_slot_pos(int32_t s) const { return seqs_[s].pos; }
    bool    seq_is_regular(int32_t s) const;
    int32_t n_free_blocks() const { return n_free_; }
    int32_t refcount(int32_t b) const { return b >= 0 && b < (int32_t) ref_.size() ? ref_[b] : -1; }
    bool    has_shift() const { return has_shift_; }
    bool    valid(int32_t s) const { return s >= 0 && s < (int32_t) seqs_.size(); }
    const pkv_cache_params & params() const { return p_; }

    // session state (/ref/src/llama-kv-cache.cpp:848-1103): the cell list a session file carries, and its restore.
    // state_cells: seq_id >= 0 -> that sequence's cells in logical order (seq_ids left empty, as the reference writes
    // n_seq_id = 0); seq_id == -1 -> every occupied physical cell once, ascending (block, offset) = the flat cell order
    // of the reference, each with the set of sequences that hold it.  PKV_ERR_INVALID while device work is pending.
    struct StateCell   { int32_t block, off, pos; std::vector<int32_t> seq_ids; };
    struct StateTarget { int32_t cell, block, off; };   // file cell -> pool slot that must receive its K/V rows
    int state_cells(int32_t seq_id, std::vector<StateCell> & out) const;
    // state_read_meta (:1003-1103): dest >= 0 replaces that sequence by the file's cells (find_slot); dest == -1 clears
    // the cache and rebuilds every sequence; blocks that hold the same file cells in the same order are shared again.
    int state_restore(int32_t dest_seq_id, const std::vector<int32_t> & pos, const std::vector<std::vector<int32_t>> & seq_ids,
                      std::vector<StateTarget> & targets);

    // llama_kv_cache_view_update (:1364-1433) over the physical cells
    void view(int32_t n_seq_max, int32_t * cell_pos, int32_t * cell_seqs, int32_t * stats) const;

    // work queued for the device (consumed by pkv_cache_update)
    std::vector<PendingOp> pending;
    // cells with a pending K-shift: (block, off, delta); clears the per-cell deltas
    void take_shifts(std::vector<int32_t> & blk, std::vector<int32_t> & off, std::vector<int32_t> & dlt);

  private:
    struct Seq {
        std::vector<int32_t> blocks;  // logical block -> physical block
        std::vector<int32_t> pos;     // per logical cell; -1 = empty (hole)
        std::vector<int32_t> delta;   // accumulated, not yet applied position shift (llama_kv_cell::delta)
    };
    std::vector<char> pinned_;                       // find_slot with evict_when_full: blocks that hold tokens of the ubatch being placed
    int32_t alloc_block(bool allow_evict = true);   // fork BlockManager policy: lowest free index first, then (evict_when_full) the lowest allocated
    void    unref_block(int32_t b);
    void    trim(Seq & s);            // drop trailing holes, wholly-empty blocks, release their blocks
    void    drop_seq(Seq & s);
    bool    cow_block(Seq & s, size_t lb, bool allow_evict = true);  // make logical block lb private to s (copy-on-write)

    pkv_cache_params     p_;
    std::vector<Seq>     seqs_;
    std::vector<int32_t> ref_;        // per physical block reference count, 0 = free
    int32_t              n_free_;
    bool                 has_shift_ = false;
};

}  // namespace pkv
