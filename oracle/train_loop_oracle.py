"""CPU ORACLE (test infrastructure, NOT product code): the trainer side of the hot path.

Restates, with paths relative to /root/reference:
    train_PPO.train control flow      algos/multiagent/train.py:259-627   -> train_loop_trace()
    AgentPPO.update_rada2c loss       algos/multiagent/ppo.py:1150-1281   -> rada2c_loss()
Only tests/ may import it.

PINNED: tests/test_train_loop_golden.py replays tests/golden/train_trace.json -- the event trace of the reference's
own train() method run over the reference's own env with recording agent stand-ins (make_golden.py gen_train_trace)
-- and tests/golden/rada2c_loss.npz -- loss / statistics / gradients / post-Adam parameters produced by the reference's
own update_rada2c with the FF_core network behind it (gen_loss).
"""
import numpy as np


def train_loop_trace(env, agent_step, number_of_agents, global_critic, steps_per_epoch, steps_per_episode, total_epochs,
                     arch="cnn"):
    """Event trace of train_PPO.train.  arch='cnn': RAD-TEAM branch.  arch='mlp'/'rnn': the RAD-A2C branch, which
    standardises observation[0] IN PLACE with a per-episode StatisticStandardization (train.py:306-311, :334-341,
    :437-441, :462-471, :504-509, :543-548) -- the branch the 2x64 MLP ('ff') collector of the build follows.

    env: an object with reset()/step(dict)/epoch_end/src (oracle.radsearch_oracle.RadSearchOracle).
    agent_step(id, observations) -> (action, state_value, logp): what ac.step returns (:349-357, :476-480).
    Returns the list of events in the order train() produces them (same vocabulary as the golden file)."""
    from oracle.radsearch_oracle import WelfordOracle
    A = number_of_agents
    ev = []
    a2c = arch in ("mlp", "rnn")
    stat = {i: WelfordOracle() for i in range(A)} if a2c else {}

    def standardize_in_place(observations):
        for i in range(A):
            o = list(observations[i])
            o[0] = stat[i].standardize(o[0])
            observations[i] = o
    f64 = lambda o: np.asarray(o, dtype=np.float64).tolist()

    def do_reset():
        ev.append(["env_reset"])
        return env.reset()[0]

    def do_agent(i, observations):
        act, val, logp = agent_step(i, observations)
        ev.append(["agent_step", i, {str(k): f64(v) for k, v in observations.items()}, int(act), float(val)])
        return act, val, logp

    observations = do_reset()                                                       # :273-281
    src = [float(env.src[0]), float(env.src[1])]                                    # :284-286
    episode_return = {i: 0.0 for i in range(A)}
    steps_in_episode = 0
    oob_count = {i: 0 for i in range(A)}
    terminal_counter = {i: 0 for i in range(A)}
    episode_count = 0
    if a2c:                                                                         # :306-311
        for i in range(A):
            stat[i].update(observations[i][0])
    for epoch in range(total_epochs):
        if a2c:                                                                     # :322-330
            for i in range(A):
                ev.append(["reset_hidden", i])
        for steps_in_epoch in range(steps_per_epoch):
            if a2c:                                                                 # :334-341
                standardize_in_place(observations)
            thoughts = {i: do_agent(i, observations) for i in range(A)}            # :347-357
            actions = {i: int(thoughts[i][0]) for i in range(A)}
            ev.append(["env_step", {str(k): v for k, v in actions.items()}])
            next_observations, rewards, terminals, infos = env.step(actions)       # :367-369
            for i in range(A):                                                      # :372-383 (float32 accumulation)
                r = rewards["individual_reward"][i] if not global_critic else rewards["team_reward"]
                episode_return[i] += float(np.array(r, dtype="float32").item())
            steps_in_episode += 1
            for i in range(A):                                                      # :386-391
                if infos[i]["out_of_bounds"]:
                    oob_count[i] += 1
            terminal_reached = False                                                # :394-398
            for i in range(A):
                if terminals[i]:
                    terminal_counter[i] += 1
                    terminal_reached = True
            timeout = steps_in_episode == steps_per_episode                         # :401-412
            episode_over = terminal_reached or timeout
            epoch_ended = steps_in_epoch == steps_per_epoch - 1
            reset_next = episode_over or epoch_ended
            for i in range(A):                                                      # :415-435
                r = rewards["individual_reward"][i] if not global_critic else rewards["team_reward"]
                ev.append(["store", i, f64(observations[i]), float(r), actions[i], float(thoughts[i][1]), float(thoughts[i][2]),
                           [float(np.float32(src[0])), float(np.float32(src[1]))], bool(reset_next)])
                if a2c:                                                             # :437-441 (inside the agent loop)
                    stat[i].update(next_observations[i][0])
            observations = next_observations                                        # :446-449
            if reset_next:                                                          # :453
                episode_count += 1
                if timeout or epoch_ended:                                          # :466-484
                    if a2c:
                        standardize_in_place(observations)
                    last_val = [do_agent(i, observations)[1] for i in range(A)]
                    if epoch_ended:
                        ev.append(["epoch_end_set", True])
                        env.epoch_end = True
                else:
                    last_val = [0 for _ in range(A)]                                # :487
                for i in range(A):
                    ev.append(["gae", i, float(last_val[i])])                       # :490-491
                if episode_over:                                                    # :494-501
                    for i in range(A):
                        ev.append(["log", i, "EpRet", float(episode_return[i])])
                        ev.append(["log", i, "EpLen", float(steps_in_episode)])
                        ev.append(["ep_len", i, steps_in_episode])
                if a2c:                                                             # :504-509
                    for i in range(A):
                        stat[i].reset()
                if a2c and not epoch_ended:                                         # :512-517
                    for i in range(A):
                        ev.append(["reset_hidden", i])
                if epoch_ended:                                                     # :519-526
                    for i in range(A):
                        ev.append(["log", i, "DoneCount", float(terminal_counter[i])])
                        ev.append(["log", i, "OutOfBound", float(oob_count[i])])
                        terminal_counter[i] = 0
                        oob_count[i] = 0
                observations = do_reset()                                           # :529-534
                src = [float(env.src[0]), float(env.src[1])]
                episode_return = {i: 0 for i in range(A)}
                steps_in_episode = 0
                if arch == "cnn":
                    for i in range(A):
                        ev.append(["reset_agent", i])                               # :537-540
                if a2c:                                                             # :543-548
                    for i in range(A):
                        stat[i].update(observations[i][0])
        if (epoch % 500 == 0) or (epoch == total_epochs - 1):                      # :552-561 (save_freq default)
            for i in range(A):
                ev.append(["save", i])
        if epoch > 99:                                                              # :564-566
            for i in range(A):
                ev.append(["reduce_pfgru", i])
        for i in range(A):                                                          # :569
            ev.append(["update", i])
        for i in range(A):                                                          # :605-627
            ev.append(["tabular", i, "TotalEnvInteracts", float((epoch + 1) * steps_per_epoch)])
            ev.append(["dump", i])
    return ev, episode_count


def rada2c_loss(logits_fn, value_fn, episodes, clip_ratio, alpha, vf_coef=0.01):
    """update_rada2c's loss and statistics (ppo.py:1191-1240) on episodes in the reference's column layout
    (obs 0:11 | adv 11 | ret 12 | logp_old 13 | act 14 | src 15:17).  logits_fn/value_fn are torch callables.
    Returns (loss tensor with graph, dict of float statistics).  The entropy term is a detached Python float in the
    reference (:1216 `.detach().mean().item()`): it moves the loss value, never the gradient."""
    import torch
    losses, kls, ents, cfs, vls = [], [], [], [], []
    for ep in episodes:
        obs, adv, ret, logp_old, act = ep[:, :11], ep[:, 11], ep[:, 12, None], ep[:, 13], ep[:, 14]
        logits = logits_fn(obs)
        logp_all = torch.log_softmax(logits, dim=-1)
        logp = logp_all.gather(1, act.long().unsqueeze(1)).squeeze(1)
        val = value_fn(obs)
        ratio = torch.exp(logp - logp_old)
        clip_adv = torch.clamp(ratio, 1 - clip_ratio, 1 + clip_ratio) * adv
        clipped = ratio.gt(1 + clip_ratio) | ratio.lt(1 - clip_ratio)
        ent = float((-(logp_all.exp() * logp_all).sum(-1)).detach().mean().item())
        val_loss = ((val - ret) ** 2).mean()
        losses.append(-(torch.min(ratio * adv, clip_adv).mean() - vf_coef * val_loss + alpha * ent))
        kls.append((logp_old - logp).detach().mean().item())
        ents.append(ent)
        cfs.append(clipped.float().mean().item())
        vls.append(val_loss.detach().item())
    loss = torch.stack(losses).mean()
    mean32 = lambda xs: float(np.mean(np.asarray(xs, dtype=np.float32)))
    return loss, dict(kl=mean32(kls), ent=mean32(ents), cf=mean32(cfs), val_loss=mean32(vls))
