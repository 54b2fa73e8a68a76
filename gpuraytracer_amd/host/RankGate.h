#pragma once
// A meeting point for the host threads of a device group (one thread per GPU, rt_headless --ranks N), placed before
// every collective call: RCCL's collectives block until every rank has called them, so a rank that has failed before one
// must not leave the others waiting inside it. Every rank arrives at the gate with its state; the gate opens when all
// ranks that are still running have arrived, and tells each of them whether all are well. A rank that gives up
// (Leave) counts as arrived, and as failed, at this and every later gate. No counterpart in the reference, which
// renders on one device.

#include <condition_variable>
#include <mutex>

namespace RT
{
	class RankGate
	{
	public:
		explicit RankGate(int iWorld) : m_iWorld(iWorld) {}

		//waits for the other ranks; true if every rank of the group is still running and arrived well
		bool Arrive(bool bWell = true)
		{
			std::unique_lock<std::mutex> stdLock(m_stdMutex);
			if (!bWell) m_bFailed = true;
			const unsigned iPhase = m_iPhase;
			if (++m_iArrived + m_iLeft >= m_iWorld) Open();
			else m_stdOpened.wait(stdLock, [&]() { return m_iPhase != iPhase; });
			//what held when this gate opened (a rank that is quicker may be at the next gate already, with news of its own;
			//the gate after the next cannot open before this rank has arrived at the next, so two slots are enough)
			return m_bWellAtOpening[iPhase & 1u];
		}

		//this rank stops taking part: the others are told at their next (or current) gate
		void Leave()
		{
			std::unique_lock<std::mutex> stdLock(m_stdMutex);
			m_bFailed = true;
			m_iLeft++;
			if (m_iArrived > 0 && m_iArrived + m_iLeft >= m_iWorld) Open();
		}

		bool Failed()
		{
			std::unique_lock<std::mutex> stdLock(m_stdMutex);
			return m_bFailed;
		}

	private:
		void Open() { m_bWellAtOpening[m_iPhase & 1u] = !m_bFailed; m_iArrived = 0; m_iPhase++; m_stdOpened.notify_all(); }

		std::mutex m_stdMutex;
		std::condition_variable m_stdOpened;
		int m_iWorld;
		int m_iArrived = 0, m_iLeft = 0;
		unsigned m_iPhase = 0;
		bool m_bWellAtOpening[2] = { true, true };
		bool m_bFailed = false;		//stays set: a group with a failed rank does no further collective
	};
}
