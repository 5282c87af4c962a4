// emgpu_term_endstep.h -- a CODE FRAGMENT of k_terminal_propagate (see emgpu_term_events.h): the end of a step whose draw was accepted --
// the turn towards the new heading (createEncounter.m:241-262), the clock, the stop conditions (CheckTrajectoryConditions, :296-329).
            att = 0;
            // ---- the step ends: turn towards the new heading, advance the clock, stop conditions
            const double turn1 = round((heading_deg - curr_hdg) * t_k(100.0)) * t_k(0.01);
            const double delta = fmin(fabs(turn1), T_LIMS(2)) * t_sign(turn1);

            // v = rotationmatrix(delta) * v  (:262; rotationmatrix(0) is the identity).  Round 4 marked the velocity "due" and the next step
            // evaluated sincosd(vang) -- 75 vector instructions for the five lanes per wave-iteration that turn, in 85 % of the wave-iterations.
            // A turn is at most maxTurnRate (<= 12 degrees in getDynamicLimits.m:15-62): cosd / sind of such an angle are the reduced-argument
            // sums themselves (n = round(delta / 90) = 0) and need a third of the terms (|x| <= 0.22: the first term left out is
            // x^13 / 13! = 4e-19 for the sine and x^12 / 12! = 2.4e-17 for the cosine -- a quarter of an ulp of a value near 1, so the sum may differ from
            // the full series' in the last bit; the GPU tests hold the tracks to the oracle's f64 rounded to f32, or one f32 step), and the rotation is the reference's own 2 x 2 product (vector instructions per launch -8 %).
            if (delta != 0.0) {
                vang += delta;
                if (fabs(delta) <= 12.5) {
                    const double x = delta * t_k(3.14159265358979323846 / 180.0), z = x * x;
                    double ps = t_k(-1.0 / 39916800.0);
                    ps = fma(ps, z, t_k(1.0 / 362880.0));
                    ps = fma(ps, z, t_k(-1.0 / 5040.0));
                    ps = fma(ps, z, t_k(1.0 / 120.0));
                    ps = fma(ps, z, t_k(-1.0 / 6.0));
                    const double sd = fma(x * z, ps, x);
                    double pc = t_k(-1.0 / 3628800.0);
                    pc = fma(pc, z, t_k(1.0 / 40320.0));
                    pc = fma(pc, z, t_k(-1.0 / 720.0));
                    pc = fma(pc, z, t_k(1.0 / 24.0));
                    const double cd = (1.0 - 0.5 * z) + (z * z) * pc;
                    const double n0 = cd * v0 - sd * v1, n1 = sd * v0 + cd * v1;
                    v0 = n0; v1 = n1;
                } else vdirty = true;   // (a limit above 12.5 degrees per second: the full evaluation at the next step, as before)
            }
            ii++;
            const double d2_nm = xy0 * xy0 + xy1 * xy1;   // (the position has not moved since the step began: recomputed, not carried)
            done = ((double)(ii - 1) > A.tmax_s) || (d2_nm > dist_hi2) || ((intent == 1 || intent == 2) && d2_nm <= 0.0625) || (ac == 0 && xy1 > 0.25);
