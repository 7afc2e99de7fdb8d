"""The reference's batch self-play harness, restated for tests that cannot import it.

`Game.batch_self_play` (src/game.py:65-164) with `AlphaZeroPlayer.get_batch_action`
(src/player.py:333-375) in the reference drive `BatchedMCTS` and the `Env` objects; fixture G8 holds
what they returned on the compiled reference.  On the GPU box the reference does not exist, so this
module restates the two functions' behaviour - which positions are searched, how moves are drawn
from numpy's global generator, what is recorded per ply, how a finished game becomes `play_data` -
on top of any object with the wrapper's interface.  tests/test_boundary_cpu.py checks the
restatement against G8 (oracle as the native backend) next to the reference's own files; the GPU
suite then runs it on the HIP engine.  Connect4 targets only (moves-left auxiliary).
"""
import numpy as np


def _temperature(ply, t0, cutoff, t_end):                       # game.py:55-63
    return t0 if (cutoff <= 0 or ply < cutoff) else t_end


def _choose(visits, temps, n_actions):
    """player.py:348-371: visit distribution and the move of every tree; trees without visits play 0."""
    acts, dists = [], []
    for v, temp in zip(visits, temps):
        dist = np.zeros(n_actions, dtype=np.float32)
        seen = v > 0
        if not seen.any():
            acts.append(0); dists.append(dist)
            continue
        dist[seen] = v[seen] / v[seen].sum()
        if temp <= 1e-6:
            a = np.argmax(v)
        else:
            logits = np.log(v[seen]) / temp
            p = np.exp(logits - np.max(logits))
            a = np.random.choice(np.where(seen)[0], p=p / np.sum(p))
        acts.append(a); dists.append(dist)
    return acts, np.array(dists)


def batch_self_play(w, pv, Env, n_games, temperature, temp_decay_moves, temp_endgame=0, td_steps=0, vl_batch=1):
    """`w`: the search wrapper (src/MCTS_cpp.py BatchedMCTS of n_games trees); returns the reference's
    list of (winner, play_data)."""
    envs = [Env() for _ in range(n_games)]
    for i in range(n_games):
        w.reset_env(i)
    log = [dict(state=[], prob=[], wdl=[], mask=[], who=[]) for _ in range(n_games)]
    running = list(range(n_games))
    finished = [None] * n_games
    while running:
        boards = np.array([e.board for e in envs])
        turns = np.array([e.turn for e in envs], dtype=np.int32)
        temps = [_temperature(len(log[i]["who"]), temperature, temp_decay_moves, temp_endgame) for i in range(n_games)]
        w.batch_playout(pv, boards, turns, vl_batch=vl_batch)
        visits = w.get_visits_count()
        rs = w.get_root_stats()
        root_wdl = np.stack([rs["root_D"], rs["root_P1W"], rs["root_P2W"]], axis=1)
        acts, dists = _choose(visits, temps, w.action_size)
        w.prune_roots(np.array(acts, dtype=np.int32))
        still = []
        for i in running:
            e, rec = envs[i], log[i]
            rec["state"].append(e.current_state()[0].astype(np.int8))
            rec["prob"].append(dists[i]); rec["wdl"].append(root_wdl[i])
            rec["mask"].append(np.array(e.valid_mask(), dtype=np.bool_)); rec["who"].append(e.turn)
            e.step(acts[i])
            if not e.done():
                still.append(i)
                continue
            winner = e.winPlayer()
            T = len(rec["who"])
            z = np.full(T, winner, dtype=np.int32)
            left = np.arange(T, 0, -1, dtype=np.int32)
            none = np.zeros(3, dtype=np.float32)
            cols = [rec["state"], rec["prob"], z, left, left, rec["wdl"], rec["mask"]]
            if td_steps > 0:
                cols.append([rec["wdl"][t + td_steps] if t + td_steps < T else none for t in range(T)])
            play = list(zip(*cols))
            last = [e.current_state()[0].astype(np.int8), np.zeros_like(rec["prob"][0]), winner, 0, 0, none,
                    np.ones_like(rec["mask"][0])]
            if td_steps > 0:
                last.append(none)
            play.append(tuple(last))
            finished[i] = (winner, tuple(play))
            w.reset_env(i)
        running = still
    return finished


def check_against_g8(data, g, bits):
    for i, (winner, play) in enumerate(data):
        assert winner == int(g[f"g{i}_winner"][0])
        for j, nm in enumerate(("state", "prob", "z", "steps", "aux", "root_wdl", "mask", "fut")):
            got = np.array([np.asarray(t[j]) for t in play])
            ref = g[f"g{i}_{nm}"]
            assert got.shape == ref.shape, (i, nm)
            if ref.dtype.kind == "f":
                assert np.array_equal(bits(got), bits(ref)), (i, nm)
            else:
                assert np.array_equal(got, ref), (i, nm)
