// emgpu_term_events.h -- a CODE FRAGMENT of k_terminal_propagate (emgpu_kernels_term.hip includes it inside the kernel's step loop, once in
// the straight-line step and once in the event-queue variant): the events of one attempt (createEncounter.m:196-243) applied to the lane's
// state.  In: dH, dA, dS (the drawn 1-based bins of heading, altitude, speed), spare (the fourth word of the attempt's TERM_TRANS block),
// st[], ac, ii, rng; out: resample ("the step is drawn again"), heading_deg / z_ft / speed / vang / vdirty / pend updated.

            // The step's dediscretize draws (slot map, round 5).  The FIRST one an attempt makes is the fourth word of the TERM_TRANS block the
            // attempt has in hand (words 0-2 are the three transition draws); only a lane with a SECOND one -- two events in one step, 0.5 % of
            // the steps -- calls Philox again (TERM_DEDISC, the variable's own word, as before).  Round 4 made that second call for every lane
            // with an event: 93 % of the wave-iterations ran it (measured without it: -4.3 %).
            const bool evH = dH != st[3] + 1, evA = dA != st[4] + 1, evS = dS != st[5] + 1;
            // MATLAB: 1:[] is empty, so with no boundary at or below the limit no altitude event is valid; []:1:e likewise for the speed
            const int alt_last = U(alt_last[ac]);
            const bool okA = evA && alt_last >= 1 && dA >= 1 && dA <= alt_last;
            resample = evA && !okA;
            const int spd_first = U(spd_first[ac]), spd_last = U(spd_last[ac]);
            const bool tryS = !resample && evS;
            const bool okS = tryS && spd_first >= 1 && dS >= spd_first && dS <= spd_last;
            resample = resample || (tryS && !okS);
            uint4 dw = make_uint4(0u, 0u, 0u, 0u);
            if ((int)evH + (int)okA + (int)okS >= 2) {
                { uint32_t k0 = (uint32_t)A.seed, k1 = (uint32_t)(A.seed >> 32); asm volatile("" : "+s"(k0), "+s"(k1)); rng.k0 = k0; rng.k1 = k1; }
                dw = rng.block(12u /* TERM_DEDISC */, 0u, (uint32_t)ii);

            }
            if (evH) {
                heading_deg = t_dedisc(s_bnd + 2 * kBndStride, dH, spare);
                const int b = t_in_bin(s_bnd + 2 * kBndStride, (int)P.i_nb[3], dH, heading_deg) ? dH : t_discretize(heading_deg, s_bnd, s_grid[2]);
                pend = (pend & 0xFFFFFF00u) | (uint32_t)(b - 1);
            }
            if (okA) {
                z_ft = t_dedisc(s_bnd + 3 * kBndStride, dA, evH ? word_of(dw, ka == 0 ? drow[0] : (ka == 1 ? drow[1] : drow[2])) : spare);
                const int b = t_in_bin(s_bnd + 3 * kBndStride, (int)P.i_nb[4], dA, z_ft) ? dA : t_discretize(z_ft, s_bnd, s_grid[3]);
                pend = (pend & 0xFFFF00FFu) | ((uint32_t)(b - 1) << 8);
            }
            if (okS) {
                double s1 = t_dedisc(s_bnd + 4 * kBndStride, dS, (evH || okA) ? word_of(dw, ks == 0 ? drow[0] : (ks == 1 ? drow[1] : drow[2])) : spare);
                const double minVel = T_LIM(0), maxVel = T_LIM(1);
                const bool inside = !(s1 < minVel) && !(s1 > maxVel) && t_in_bin(s_bnd + 4 * kBndStride, (int)P.i_nb[5], dS, s1);
                if (s1 < minVel) s1 = minVel;
                if (s1 > maxVel) s1 = maxVel;
                const int b = inside ? dS : t_discretize(s1, s_bnd, s_grid[4]);   // (a clamped speed may have left its bin)
                pend = (pend & 0x0000FFFFu) | ((uint32_t)(b - 1) << 16);
                vang = heading_deg; vdirty = true;     // v = rotationmatrix(heading_deg) * [s1; 0]  (:246-247)
                speed = s1;
            }
