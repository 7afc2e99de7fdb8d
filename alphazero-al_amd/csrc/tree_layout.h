// tree_layout.h - how search trees live in HBM.
//
// The reference keeps one AoS pool per tree: 64-byte nodes that point to 16-byte edges that
// point back to lazily allocated child nodes (MCTSNode.h:69-140, 149-199), so one PUCT step
// chases parent -> edges -> up to 7 scattered children.  Here a node's children are ONE
// contiguous block of records allocated at expansion time; a record carries both the edge
// payload (action, prior, noise) and the child's statistics.  One PUCT step is then a single
// coalesced load of E adjacent 32-byte records by E adjacent lanes.  "Lazy child allocation"
// (MCTS.h:268-275) becomes setting the EXISTS bit of a record that is already there; node
// numbering is not observable through the API, so this does not change any result.
//
//   hot[tree*S + slot]   32 B  everything selection reads
//   cold[tree*S + slot]  16 B  what only backup / root queries / root noise read
//
// A tree's arena is TWO HALVES of S records, and the tree lives in one of them at a time, in slots
// [0, used); slot 0 is its root (a freshly reset root, or the root a re-rooting put there; it has
// no parent block; parent == -1 marks it, MCTS.h:100-101).  The reference never reclaims a node
// before the next reset (MCTS.h:90-108: its pools only grow), so over a game a tree's pool holds
// every node ever expanded - n_playout x actions x plies records in the worst case.  Here a
// re-rooting (k_prune) copies the subtree it keeps into the other half, breadth first, and flips
// the tree over: a tree occupies what is reachable from its root, and S is sized for one ply's
// growth on top of that instead of for a whole game.
#pragma once

#include <cstdint>

namespace az {

struct alignas(16) HotRec {
    int32_t  n_visits;    // real visits N                      (MCTSNode.h:92)
    int32_t  n_inflight;  // virtual-loss visits in flight      (MCTSNode.h:93)
    float    w_p1;        // sum of P1-win probability, absolute view (MCTSNode.h:89-90)
    float    w_p2;        // sum of P2-win probability
    float    m_sum;       // sum of moves-left                  (MCTSNode.h:95)
    float    prior;       // P(a) of the edge leading here      (MCTSNode.h:73)
    int32_t  child_off;   // first slot of this node's child block (edge_offset), -1 if none
    uint32_t meta;        // packed, see below
};
static_assert(sizeof(HotRec) == 32, "hot record is two dwordx4");

struct alignas(16) ColdRec {
    float   w_draw;       // sum of draw probability            (MCTSNode.h:88)
    float   noise;        // Dirichlet noise of the edge leading here (MCTSNode.h:74)
    int32_t parent;       // slot of the parent node, -1 for the root (MCTSNode.h:102)
    int32_t reserved;
};
static_assert(sizeof(ColdRec) == 16, "cold record is one dwordx4");

// HotRec::meta
constexpr uint32_t META_ACTION_MASK = 0xffu;       // action of the edge leading here
constexpr int      META_NEDGE_SHIFT = 8;           // number of children (num_edges), 8 bits
constexpr uint32_t META_NEDGE_MASK  = 0xffu << META_NEDGE_SHIFT;
constexpr uint32_t META_TURN_P1     = 1u << 16;    // side to move at this node is +1 (node.turn == 1)
constexpr uint32_t META_EXPANDED    = 1u << 17;    // is_expanded
constexpr uint32_t META_TERMINAL    = 1u << 18;    // is_terminal
constexpr int      META_RESULT_SHIFT = 19;         // cached terminal result: 0 draw, 1 P1 wins, 2 P2 wins
constexpr uint32_t META_RESULT_MASK = 3u << META_RESULT_SHIFT;
constexpr uint32_t META_EXISTS      = 1u << 21;    // the reference would have allocated this child

// per-leaf flags produced by selection
constexpr uint8_t LEAF_TERMINAL   = 1u << 0;
constexpr int     LEAF_RESULT_SHIFT = 1;           // 2 bits, same coding as META_RESULT
constexpr uint8_t LEAF_VL_APPLIED = 1u << 3;       // the descent left in-flight visits on its path
constexpr uint8_t LEAF_ROOT_UNEXPANDED = 1u << 4;  // leaf is the (unexpanded) root: expansion draws noise
constexpr uint8_t LEAF_EXPANDED   = 1u << 5;       // leaf was already expanded when selected

// Connect4 geometry (Connect4.h:37-46)
constexpr int C4_ROWS = 6, C4_COLS = 7, C4_CELLS = 42, C4_ACTIONS = 7, C4_BITS_PER_COL = 7;
constexpr int C4_MAX_PATH = 44;                    // root + at most 42 plies, padded
constexpr int C4_STATS = 6 + 8 * C4_ACTIONS;       // get_root_stats row (MCTS.h:634-635)

// Othello geometry (Othello.h:35-46): 64 squares + pass, up to 33 legal moves, a descent is at
// most 60 placements plus passes
constexpr int OT_CELLS = 64, OT_ACTIONS = 65, OT_MAX_PATH = 136;
constexpr int OT_STATS = 6 + 8 * OT_ACTIONS;

constexpr int LANES_PER_TREE = 8;                  // Connect4: one lane per edge (7 used), 8 trees per wave

}  // namespace az
